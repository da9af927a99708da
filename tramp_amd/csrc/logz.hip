// logz.hip - log-partitions of the hot-path factors and of the variables:
// the terms of the EP log-evidence (tramp/algos/message_passing.py:306-328,
// tramp/algos/expectation_propagation.py:21-27).  One streaming reduction per call: every
// component's log-partition is formed in double precision (the sums cancel between factors
// and variables) and summed per instance in a fixed order; the host mirror divides by n where
// the reference takes the mean (priors, likelihoods) and keeps the sum where it sums
// (channels, variables) - the reference's own mixed convention.
#include "common.h"
#include "linear.h"
#include "tramp_math.h"

namespace {

constexpr double k2Pi = 2 * tmath::kPi;

__device__ __forceinline__ double normal_A(double a, double b) {            // beliefs/normal.py:3-4
    return 0.5 * (b * b / a + log(k2Pi / a));
}

__device__ __forceinline__ double logaddexp(double x, double y) {            // np.logaddexp
    const double m = fmax(x, y);
    if (!(m > -INFINITY)) return m;
    return m + log1p(exp(-fabs(x - y)));
}

struct LogzParams {
    int kind;
    double p0, p1, p2;
    const double* a0;
    const double* a1;
};

// truncated_normal_logZ of N(r0, v0) on the half line (tramp/utils/truncated_normal.py:294-298)
__device__ __forceinline__ double halfline_logZ(double r0, double v0, int side) {
    tmath::HalfLine<double> h = tmath::halfline<double>(r0, v0, side);
    return 0.5 * log(k2Pi * v0) + 0.5 * r0 * r0 / v0 + h.logp;
}

template <int KIND>
__device__ __forceinline__ double pwl_logZ(double bz, double bx, double az, double ax) {
    double As[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {                     // linear_region.py:60-64
        const double slope = tmath::pwl_slope(KIND, k), x0 = tmath::pwl_x0(KIND, k);
        const double a = az + slope * slope * ax;
        const double b = bz + slope * (bx - ax * x0);
        As[k] = halfline_logZ(b / a, 1.0 / a, k == 0 ? +1 : -1) - 0.5 * ax * x0 * x0 + bx * x0;
    }
    return logaddexp(As[0], As[1]);                   // piecewise_linear_channel.py:55-58
}

// ---- region tables: the rest of the piecewise-linear family (LeakyRelu, AsymmetricAbs, HardTanh, HardSigmoid,
// SymmetricDoor and the likelihood twins), literally as the reference loops over its regions
struct RegionTable {
    int nreg;
    double zmin[3], zmax[3], slope[3], x0[3];
};

// truncated_normal_logZ (tramp/utils/truncated_normal.py:294-298) through truncated_normal_log_proba (:269-281):
// G0_inf = log_Phi with its x < 30 mask on a half line, G0 = log(1/2) + F0 on a finite interval
__device__ __forceinline__ double trunc_logZ(double r0, double v0, double zmin, double zmax) {
    double g0;
    if (zmax == INFINITY) g0 = tmath::halfline_at<double>(r0, v0, +1, zmin).logp;
    else if (zmin == -INFINITY) g0 = tmath::halfline_at<double>(r0, v0, -1, zmax).logp;
    else g0 = tmath::interval_moments<double>(r0, v0, zmin, zmax).logp;
    return 0.5 * log(k2Pi * v0) + 0.5 * r0 * r0 / v0 + g0;
}

// scipy.special.logsumexp over <= 3 terms: the maximum is taken out unless it is not finite
__device__ __forceinline__ double logsumexp3(const double* A, int n) {
    double m = A[0];
    for (int k = 1; k < n; ++k) m = (A[k] > m || m != m) ? A[k] : m;          // np.max: NaN wins
    for (int k = 0; k < n; ++k) if (A[k] != A[k]) m = A[k];
    const double mm = (m - m == 0.0) ? m : 0.0;
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += exp(A[k] - mm);
    return log(s) + mm;
}

// mode 0: channel, v1 = bx (linear_region.py:55-64, piecewise_linear_channel.py:55-58, summed);
// mode 1: likelihood, v1 = y (piecewise_linear_likelihood.py:52-61, 216-219; the host takes the mean).
// log|slope|: the reference writes np.log(self.slope), NaN for the descending branch of AsymmetricAbsLikelihood - its
// every output is NaN; the Jacobian of y = x0 + slope z is |slope|.
template <typename T>
__global__ __launch_bounds__(kBlock) void region_logz_kernel(RegionTable R, int mode, size_t n,
                                                             const double* __restrict__ az_d,
                                                             const double* __restrict__ ax_d, const T* __restrict__ v0,
                                                             const T* __restrict__ v1, double* __restrict__ partials) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    const double az = az_d[b], ax = ax_d ? ax_d[b] : 0.0;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const double bz = (double)v0[b * n + i], w = (double)v1[b * n + i];
        double As[3];
        for (int k = 0; k < R.nreg; ++k) {
            const double slope = R.slope[k], x0 = R.x0[k];
            if (mode == 0) {
                const double a = az + slope * slope * ax, bb = bz + slope * (w - ax * x0);
                As[k] = trunc_logZ(bb / a, 1.0 / a, R.zmin[k], R.zmax[k]) - 0.5 * ax * x0 * x0 + w * x0;
            } else if (slope == 0.0) {
                As[k] = (w == x0) ? trunc_logZ(bz / az, 1.0 / az, R.zmin[k], R.zmax[k]) : -INFINITY;
            } else {
                const double z = (w - x0) / slope;
                As[k] = (R.zmin[k] < z && z < R.zmax[k]) ? -0.5 * az * (z * z) + bz * z - log(fabs(slope)) : -INFINITY;
            }
        }
        acc += logsumexp3(As, R.nreg);
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = acc;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void logz_kernel(LogzParams P, size_t n, const T* __restrict__ v0,
                                                      const T* __restrict__ v1, double* __restrict__ partials) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    const double a0 = P.a0 ? P.a0[b] : 0.0, a1 = P.a1 ? P.a1[b] : 0.0;
    const T* p0 = v0 + b * n;
    const T* p1 = v1 ? v1 + b * n : nullptr;
    double acc = 0.0;
    // per-instance constants
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    switch (P.kind) {
        case TRAMP_LOGZ_GB_PRIOR: {                   // p0 rho, p1 mean, p2 var
            c0 = 1.0 / P.p2;                          // a0
            c1 = P.p1 / P.p2;                         // b0
            c2 = normal_A(c0, c1) - log(P.p0 / (1 - P.p0));   // eta (gauss_bernoulli_prior.py:32-36)
            c3 = logaddexp(c2, normal_A(c0, c1));
            break;
        }
        case TRAMP_LOGZ_GAUSS_PRIOR:                  // p0 mean, p1 var
            c0 = 1.0 / P.p1;
            c1 = P.p0 / P.p1;
            c3 = normal_A(c0, c1);
            break;
        case TRAMP_LOGZ_GAUSS_LIK:                    // p0 var
            c0 = 1.0 / P.p0;
            break;
        default:
            break;
    }
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const double x = (double)p0[i];
        const double y = p1 ? (double)p1[i] : 0.0;
        double A;
        switch (P.kind) {
            case TRAMP_LOGZ_GB_PRIOR:                 // gauss_bernoulli_prior.py:79-83, sparse.A
                A = logaddexp(c2, normal_A(a0 + c0, x + c1)) - c3;
                break;
            case TRAMP_LOGZ_GAUSS_PRIOR:              // gaussian_prior.py:70-74
                A = normal_A(a0 + c0, x + c1) - c3;
                break;
            case TRAMP_LOGZ_L1_PRIOR: {               // map_L1_norm_prior.py:73-76; p0 gamma
                const double g = P.p0;
                const double sh = fabs(x) > g ? x - copysign(g, x) : 0.0;
                const double rx = sh / a0;
                A = x * rx - 0.5 * a0 * rx * rx - g * fabs(rx);
                break;
            }
            case TRAMP_LOGZ_GAUSS_LIK: {              // gaussian_likelihood.py:51-56
                const double by = c0 * y;
                A = normal_A(a0 + c0, x + by) - normal_A(c0, by);
                break;
            }
            case TRAMP_LOGZ_SGN_LIK:                  // sgn_likelihood.py:39-41, positive.A
                A = halfline_logZ((x * y) / a0, 1.0 / a0, +1);
                break;
            case TRAMP_LOGZ_ABS_LIK:                  // abs_likelihood.py:38-40, binary.A
                A = -0.5 * a0 * y * y + logaddexp(x * y, -(x * y));
                break;
            case TRAMP_LOGZ_PWL_SGN:
                A = pwl_logZ<0>(x, y, a0, a1);
                break;
            case TRAMP_LOGZ_PWL_ABS:
                A = pwl_logZ<1>(x, y, a0, a1);
                break;
            case TRAMP_LOGZ_PWL_RELU:
                A = pwl_logZ<2>(x, y, a0, a1);
                break;
            default: {                                // TRAMP_LOGZ_VARIABLE, tramp/base.py:146-150
                const double a = a0 + a1, bb = x + y;
                A = (a <= 0.0) ? INFINITY : 0.5 * (bb * bb / a + log(k2Pi / a));
                break;
            }
        }
        acc += A;
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = acc;
}

__global__ __launch_bounds__(kBlock) void logz_fold_kernel(const double* __restrict__ partials, int nblk,
                                                           double* __restrict__ out) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) s += partials[b * nblk + i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[b] = s;
}

// LinearChannel.compute_log_partition (linear_channel.py:127-132) in the thin basis:
//   b . rz = sum_k (t + sing u)^2 / D + (|bz|^2 - |t|^2) / az        (the last term when k < N)
//   sum log(2 pi / (az + ax s)) = sum_k log(2 pi / D) + (N - k) log(2 pi / az)
template <typename T>
__global__ __launch_bounds__(kBlock) void lin_logz_kernel(int k, size_t N, const double* __restrict__ sing,
                                                          const double* __restrict__ az_d,
                                                          const double* __restrict__ ax_d, const T* __restrict__ t,
                                                          const T* __restrict__ u, const T* __restrict__ bz,
                                                          double* __restrict__ out) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.x;
    const double az = az_d[b], ax = ax_d[b];
    double acc = 0.0;
    for (int i = threadIdx.x; i < k; i += kBlock) {
        const double s = sing[i], D = az + ax * s * s;
        const double ti = (double)t[b * k + i], c = ti + s * (double)u[b * k + i];
        acc += 0.5 * c * c / D + 0.5 * log(k2Pi / D);
        if ((size_t)k < N) acc -= 0.5 * ti * ti / az;
    }
    if ((size_t)k < N) {
        for (size_t i = threadIdx.x; i < N; i += kBlock) {
            const double v = (double)bz[b * N + i];
            acc += 0.5 * v * v / az;
        }
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) out[b] = acc + ((size_t)k < N ? 0.5 * (double)(N - k) * log(k2Pi / az) : 0.0);
}


// ---- round 6: ExponentialPrior / PositivePrior (which = 0: params {a0, b0, A0}) and GaussianMixturePrior (which = 1:
// params {K, probs[K], means[K], vars[K]}); the host takes the mean as the reference does.
//   exponential_prior.py:62-65: positive.A(ax, bx + b0) - exponential.A(b0);  positive_prior.py:65-69: positive.A(ax + 1, bx)
//   - positive.A(1, 0)  (A0 = the subtracted constant);  gaussian_mixture_prior.py:99-104 with beliefs/mixture.py:18-20
struct Prior2Params { int which, K; double a0, b0, A0; double ga0[tmath::kGmMax], gb0[tmath::kGmMax], geta[tmath::kGmMax]; };

template <typename T>
__global__ __launch_bounds__(kBlock) void prior2_logz_kernel(Prior2Params P, size_t n, const double* __restrict__ ax_d,
                                                             const T* __restrict__ bx, double* __restrict__ partials) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    const double ax = ax_d[b];
    const T* p0 = bx + b * n;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const double x = (double)p0[i];
        double A;
        if (P.which == 0) {
            const double a = ax + P.a0;
            A = halfline_logZ((x + P.b0) / a, 1.0 / a, +1) - P.A0;
        } else if (P.which == 2) {
            // BinaryPrior (binary_prior.py:48-51, 64-67; beliefs/binary.py:4-5): logaddexp(b, -b) = |b| + log1p(exp(-2 |b|))
            const double bb = fabs(x + P.b0);
            A = bb + log1p(exp(-2.0 * bb)) - P.A0 - 0.5 * ax;
        } else {
            double xi[tmath::kGmMax];
            for (int k = 0; k < P.K; ++k) xi[k] = P.geta[k] + normal_A(ax + P.ga0[k], x + P.gb0[k]);
            double m = xi[0];
            for (int k = 1; k < P.K; ++k) m = (xi[k] > m || m != m) ? xi[k] : m;
            const double mm = (m - m == 0.0) ? m : 0.0;                      // scipy.special.logsumexp
            double s = 0.0;
            for (int k = 0; k < P.K; ++k) s += exp(xi[k] - mm);
            A = log(s) + mm - P.A0;
        }
        acc += A;
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = acc;
}

}  // namespace

extern "C" int tramp_log_partition(tramp_ctx* ctx, int dtype, int kind, size_t n, size_t batch, double p0, double p1,
                                   double p2, const double* a0_d, const void* v0_d, const double* a1_d,
                                   const void* v1_d, double* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, v0_d && out_d && a0_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, kind >= TRAMP_LOGZ_GB_PRIOR && kind <= TRAMP_LOGZ_VARIABLE);
    const bool needs_v1 = kind >= TRAMP_LOGZ_GAUSS_LIK;
    TRAMP_REQUIRE(ctx, !needs_v1 || v1_d);
    const bool needs_a1 = kind >= TRAMP_LOGZ_PWL_SGN;
    TRAMP_REQUIRE(ctx, !needs_a1 || a1_d);
    if (!batch) return TRAMP_OK;
    if (!n) return tramp_fill_f64(ctx, out_d, batch, 0.0);
    const int nblk = blocks_for(n, 1);
    int rc = tramp_ws_reserve(ctx, batch * (size_t)nblk * sizeof(double));
    if (rc) return rc;
    LogzParams P{kind, p0, p1, p2, a0_d, needs_a1 ? a1_d : nullptr};
    dim3 grid(nblk, (unsigned)batch);
    ProfScope prof(ctx, PROF_OTHER, (double)(needs_v1 ? 2 : 1) * dtype_size(dtype) * (double)n * (double)batch, 0);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((logz_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, P, n, (const double*)v0_d,
                           needs_v1 ? (const double*)v1_d : nullptr, ctx->ws);
    else
        hipLaunchKernelGGL((logz_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, P, n, (const float*)v0_d,
                           needs_v1 ? (const float*)v1_d : nullptr, ctx->ws);
    hipLaunchKernelGGL(logz_fold_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, ctx->ws, nblk, out_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_prior2_log_partition(tramp_ctx* ctx, int dtype, int which, size_t n, size_t batch,
                                          const double* params_h, int nparams, const double* ax_d, const void* bx_d,
                                          double* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, params_h && ax_d && bx_d && out_d && which >= 0 && which <= 2);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    Prior2Params P{};
    P.which = which;
    if (which == 0) {
        TRAMP_REQUIRE(ctx, nparams >= 3 && params_h[0] >= 0.0);
        P.a0 = params_h[0];
        P.b0 = params_h[1];
        P.A0 = params_h[2];
    } else if (which == 2) {
        TRAMP_REQUIRE(ctx, nparams >= 1);
        P.b0 = params_h[0];                              // the prior's natural parameter, A0 = binary.A(b0)
        P.A0 = fabs(P.b0) + log1p(exp(-2.0 * fabs(P.b0)));
    } else {
        const int K = (int)params_h[0];
        TRAMP_REQUIRE(ctx, K >= 1 && K <= tmath::kGmMax && nparams >= 1 + 3 * K);
        P.K = K;
        double xi0[tmath::kGmMax], m = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const double pk = params_h[1 + k], mk = params_h[1 + K + k], vk = params_h[1 + 2 * K + k];
            TRAMP_REQUIRE(ctx, pk > 0.0 && vk > 0.0);
            P.ga0[k] = 1.0 / vk;
            P.gb0[k] = mk / vk;
            const double A = 0.5 * (P.gb0[k] * P.gb0[k] / P.ga0[k] + log(k2Pi / P.ga0[k]));
            P.geta[k] = log(pk) - A;
            xi0[k] = P.geta[k] + A;                    // = log p_k: mixture.A(a0, b0, eta) = log sum p_k
            m = xi0[k] > m ? xi0[k] : m;
        }
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += exp(xi0[k] - m);
        P.A0 = log(s) + m;
    }
    if (!batch) return TRAMP_OK;
    if (!n) return tramp_fill_f64(ctx, out_d, batch, 0.0);
    TRAMP_REQUIRE(ctx, batch <= 65535);
    const int nblk = blocks_for(n, 1);
    int rc = tramp_ws_reserve(ctx, batch * (size_t)nblk * sizeof(double));
    if (rc) return rc;
    dim3 grid(nblk, (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((prior2_logz_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, P, n, ax_d, (const double*)bx_d, ctx->ws);
    else
        hipLaunchKernelGGL((prior2_logz_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, P, n, ax_d, (const float*)bx_d, ctx->ws);
    hipLaunchKernelGGL(logz_fold_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, ctx->ws, nblk, out_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_region_log_partition(tramp_ctx* ctx, int dtype, int mode, int nreg, const double* regions_h,
                                          size_t n, size_t batch, const double* az_d, const void* bz_d,
                                          const double* ax_d, const void* v1_d, double* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, regions_h && az_d && bz_d && v1_d && out_d && (nreg == 2 || nreg == 3));
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F32 || dtype == TRAMP_F64);
    TRAMP_REQUIRE(ctx, (mode == 0 && ax_d) || mode == 1);
    RegionTable R{};
    R.nreg = nreg;
    for (int k = 0; k < nreg; ++k) {
        R.zmin[k] = regions_h[4 * k];
        R.zmax[k] = regions_h[4 * k + 1];
        R.slope[k] = regions_h[4 * k + 2];
        R.x0[k] = regions_h[4 * k + 3];
        TRAMP_REQUIRE(ctx, R.zmin[k] < R.zmax[k]);                     // linear_region.py:17
    }
    if (!batch) return TRAMP_OK;
    if (!n) return tramp_fill_f64(ctx, out_d, batch, 0.0);
    const int nblk = blocks_for(n, 1);
    int rc = tramp_ws_reserve(ctx, batch * (size_t)nblk * sizeof(double));
    if (rc) return rc;
    dim3 grid(nblk, (unsigned)batch);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((region_logz_kernel<double>), grid, dim3(kBlock), 0, ctx->stream, R, mode, n, az_d,
                           mode == 0 ? ax_d : nullptr, (const double*)bz_d, (const double*)v1_d, ctx->ws);
    else
        hipLaunchKernelGGL((region_logz_kernel<float>), grid, dim3(kBlock), 0, ctx->stream, R, mode, n, az_d,
                           mode == 0 ? ax_d : nullptr, (const float*)bz_d, (const float*)v1_d, ctx->ws);
    hipLaunchKernelGGL(logz_fold_kernel, dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, ctx->ws, nblk, out_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

// t and u of the handle must be current (call after a tramp_linear_message with the same
// messages, or pass reuse = 0 to recompute both projections here).
int lin_project(tramp_ctx* ctx, tramp_linear* lin, size_t batch, const void* bz_d, const void* bx_d, int reuse);

extern "C" int tramp_linear_log_partition(tramp_ctx* ctx, tramp_linear* lin, size_t batch, const double* az_d,
                                          const void* bz_d, const double* ax_d, const void* bx_d, int reuse,
                                          double* out_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, lin && az_d && bz_d && ax_d && bx_d && out_d);
    if (!batch) return TRAMP_OK;
    int rc = lin_project(ctx, lin, batch, bz_d, bx_d, reuse);
    if (rc) return rc;
    if (lin->dtype == TRAMP_F64)
        hipLaunchKernelGGL((lin_logz_kernel<double>), dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, (int)lin->k,
                           lin->N, lin->sing, az_d, ax_d, (const double*)lin->t, (const double*)lin->u,
                           (const double*)bz_d, out_d);
    else
        hipLaunchKernelGGL((lin_logz_kernel<float>), dim3((unsigned)batch), dim3(kBlock), 0, ctx->stream, (int)lin->k,
                           lin->N, lin->sing, az_d, ax_d, (const float*)lin->t, (const float*)lin->u,
                           (const float*)bz_d, out_d);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}
