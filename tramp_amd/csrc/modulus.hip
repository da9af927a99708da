// modulus.hip - ModulusLikelihood y = |z| for complex z (tramp/likelihoods/modulus_likelihood.py:36-131), the output
// factor of phase retrieval (models/generalized_linear_model.py:26-28, models/dag_algebra.py:38-39).
//
// A complex vector is a PLANAR real array Z[0] = Re z, Z[1] = Im z (tramp/utils/misc.py:13-27): device messages are
// [batch, 2, n], the observation y is [batch, n].  Per component
//     b = |bz|,  I = I1(b y) / I0(b y),  rz = (bz / b) y I,  v_i = y^2 (1 - I^2) / 2,   v = mean_i v_i   (:82-93)
// then compute_ab_new on the planar array (tramp/base.py:250-255).  The Bessel ratio is evaluated in double whatever the
// element type (tramp_math.h: bessel_i1_over_i0).  Three launches like the other separable factors whose update needs
// the instance mean first: moments -> scalar -> apply.  The update is bound by the Bessel ratio (two power series of
// 10 + 1.6 |bz| y terms per component), so pass 1 parks the ratio I (one double per complex component, workspace) and
// pass 2 only rebuilds the phase bz / |bz|.
#include "common.h"
#include "tramp_math.h"

namespace {

template <typename T>
__global__ __launch_bounds__(kBlock) void modulus_pass1_kernel(size_t n, const T* __restrict__ bz, const T* __restrict__ y,
                                                               double* __restrict__ partials, double* __restrict__ ratio) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    const T* re = bz + b * 2 * n;
    const T* im = re + n;
    const T* yy = y + b * n;
    double sv = 0.0;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (size_t)gridDim.x * kBlock) {
        const double yv = (double)yy[j];
        const double I = tmath::bessel_i1_over_i0(::hypot((double)re[j], (double)im[j]) * yv);
        ratio[b * n + j] = I;
        sv += 0.5 * (yv * yv) * (1.0 - I * I);
    }
    sv = block_sum(sv, smem);
    if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = sv;
}

__global__ __launch_bounds__(kBlock) void modulus_finish_kernel(const double* __restrict__ partials, int nblk, double dn,
                                                                const double* __restrict__ az, double damping,
                                                                const double* a_old, double* a_new, double* a_raw,
                                                                double* v_out) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) s += partials[b * nblk + i];
    s = block_sum(s, smem);
    if (threadIdx.x != 0) return;
    const double a = az[b];
    const double v = s / dn;
    const double an = tmath::a_new_of(v, a);
    a_raw[b] = an;
    if (a_new) a_new[b] = (damping != 0.0) ? damping * a_old[b] + (1.0 - damping) * an : an;
    if (v_out) v_out[b] = v;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void modulus_pass2_kernel(size_t n, const double* __restrict__ az,
                                                               const double* __restrict__ a_raw, const T* bz,
                                                               const T* __restrict__ y, const double* __restrict__ ratio,
                                                               double damping, const T* b_old, T* b_new, T* r_out) {
    const size_t b = blockIdx.y;
    const double asum = az[b] + a_raw[b];
    const size_t off = b * 2 * n;
    const T* yy = y + b * n;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (size_t)gridDim.x * kBlock) {
        const double re = (double)bz[off + j], im = (double)bz[off + n + j];
        const double bm = ::hypot(re, im), sc = (double)yy[j] * ratio[b * n + j];
        tmath::ModOut o;
        o.re = (bm != 0.0) ? (re / bm) * sc : 0.0;                       // tmath::modulus_posterior with I from pass 1
        o.im = (bm != 0.0) ? (im / bm) * sc : 0.0;
        double b0 = o.re * asum - re, b1 = o.im * asum - im;
        if (damping != 0.0) {
            b0 = damping * (double)b_old[off + j] + (1.0 - damping) * b0;
            b1 = damping * (double)b_old[off + n + j] + (1.0 - damping) * b1;
        }
        if (b_new) {
            b_new[off + j] = (T)b0;
            b_new[off + n + j] = (T)b1;
        }
        if (r_out) {
            r_out[off + j] = (T)o.re;
            r_out[off + n + j] = (T)o.im;
        }
    }
}

// scalar_log_partition AS FOUND (modulus_likelihood.py:125-129): np.absolute acts on the PLANAR array, i.e. on each real
// coordinate separately: A = -az y^2 / 2 + log(2 pi y ive(0, |b| y)) + |b| y summed over both planes; the host divides
// by 2n ("A.mean() / 2").  ive(0, x) = exp(-x) I0(x): log(ive(0, x)) + x = log I0(x).
__device__ __forceinline__ double log_i0(double x) {
    const double ax = fabs(x);
    if (ax <= 25.0) {
        const double t = 0.25 * ax * ax;
        double t0 = 1.0, s0 = 1.0;
        for (int k = 1; k <= 48; ++k) {
            t0 *= t / ((double)k * (double)k);
            s0 += t0;
        }
        return log(s0);
    }
    const double z = 1.0 / ax;
    double t0 = 1.0, s0 = 1.0;
    for (int k = 1; k <= 18; ++k) {
        const double q = (double)(2 * k - 1) * (double)(2 * k - 1);
        t0 *= z * (q / (8.0 * (double)k));
        s0 += t0;
    }
    return ax - 0.5 * log(2 * tmath::kPi * ax) + log(s0);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void modulus_logz_kernel(size_t n, const double* __restrict__ az,
                                                              const T* __restrict__ bz, const T* __restrict__ y,
                                                              double* __restrict__ partials) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    const double a = az[b];
    const T* yy = y + b * n;
    double acc = 0.0;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < 2 * n; j += (size_t)gridDim.x * kBlock) {
        const double yv = (double)yy[j < n ? j : j - n];
        const double bb = fabs((double)bz[b * 2 * n + j]);
        acc += -0.5 * a * yv * yv + log(2 * tmath::kPi * yv) + log_i0(bb * yv);
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = acc;
}

__global__ __launch_bounds__(kBlock) void modulus_fold_kernel(const double* __restrict__ partials, int nblk,
                                                              double* __restrict__ out) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) s += partials[b * nblk + i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[b] = s;
}

}  // namespace

extern "C" int tramp_modulus_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const double* az_d,
                                     const void* bz_d, const void* y_d, double damping, const double* a_old_d,
                                     const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d, double* v_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && bz_d && y_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!n || !batch) return TRAMP_OK;
    TRAMP_REQUIRE(ctx, batch <= 65535);
    const int nblk = blocks_for(n, 1);
    int rc = tramp_ws_reserve(ctx, batch * ((size_t)nblk + 1 + n) * sizeof(double));
    if (rc) return rc;
    double* partials = ctx->ws;
    double* a_raw = ctx->ws + batch * (size_t)nblk;
    double* ratio = a_raw + batch;
    dim3 grid(nblk, (unsigned)batch);
    // algorithmic traffic: both planes of the message and the observation in, both planes of the update out
    ProfScope prof(ctx, PROF_EW, 5.0 * dtype_size(dtype) * (double)n * (double)batch, 0.0);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((modulus_pass1_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, (const double*)bz_d,
                           (const double*)y_d, partials, ratio);
    else
        hipLaunchKernelGGL((modulus_pass1_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, (const float*)bz_d,
                           (const float*)y_d, partials, ratio);
    hipLaunchKernelGGL(modulus_finish_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, partials, nblk,
                       (double)n, az_d, damping, a_old_d, a_new_d, a_raw, v_d);
    if (b_new_d || r_d) {
        if (dtype == TRAMP_F64)
            hipLaunchKernelGGL((modulus_pass2_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, az_d, a_raw,
                               (const double*)bz_d, (const double*)y_d, ratio, damping, (const double*)b_old_d,
                               (double*)b_new_d, (double*)r_d);
        else
            hipLaunchKernelGGL((modulus_pass2_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, az_d, a_raw,
                               (const float*)bz_d, (const float*)y_d, ratio, damping, (const float*)b_old_d, (float*)b_new_d,
                               (float*)r_d);
    }
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_modulus_log_partition(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const double* az_d,
                                           const void* bz_d, const void* y_d, double* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az_d && bz_d && y_d && out_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    if (!batch) return TRAMP_OK;
    if (!n) return tramp_fill_f64(ctx, out_d, batch, 0.0);
    TRAMP_REQUIRE(ctx, batch <= 65535);
    const int nblk = blocks_for(2 * n, 1);
    int rc = tramp_ws_reserve(ctx, batch * (size_t)nblk * sizeof(double));
    if (rc) return rc;
    dim3 grid(nblk, (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((modulus_logz_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, n, az_d, (const double*)bz_d,
                           (const double*)y_d, ctx->ws);
    else
        hipLaunchKernelGGL((modulus_logz_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, n, az_d, (const float*)bz_d,
                           (const float*)y_d, ctx->ws);
    hipLaunchKernelGGL(modulus_fold_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, ctx->ws, nblk, out_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}
