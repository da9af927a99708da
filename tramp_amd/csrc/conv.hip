// conv.hip - circular-convolution linear channel x = w * z (real signals, 1-3 dimensions).
//
// tramp/channels/linear/conv_channel.py:13-149.  The operator is diagonal in the Fourier
// basis, so the factor update is O(n log n) and never touches a dense matrix: with
//     fz = rfft(bz), fx = rfft(bx), wbar = rfft(filter), s = |wbar|^2
//     rz_fft = (fz + wbar fx) / (az + ax s)          (conv_channel.py:86-97)
//     rx_fft = conj(wbar) rz_fft                     (conv_channel.py:105-109)
// Real input => only the Hermitian half spectrum (last axis n_last/2+1) is stored and
// streamed.  The transforms themselves are rocFFT plans (batched over the ensemble axis,
// executing on the context's stream with a pre-reserved work buffer); everything between
// them - the spectral solve, the message update epilogue, the scalar n_eff reduction - is
// hand-written and HBM-streaming.  Like the dense channel, the handle keeps fz and fx so
// the forward and backward update of one EP iteration share them (TRAMP_REUSE_T/U).
#include <rocfft/rocfft.h>

#include <mutex>
#include <vector>

#include "common.h"
#include "fft_pow2.h"
#include "tramp_math.h"

namespace {

struct ConvPlans {
    size_t batch;
    rocfft_plan fwd, inv;
    rocfft_execution_info info_fwd, info_inv;
    void* work_fwd;
    void* work_inv;
};

std::once_flag g_rocfft_once;

template <typename T> struct Cx;
template <> struct Cx<double> { using type = double2; };
template <> struct Cx<float> { using type = float2; };

}  // namespace

struct tramp_conv {
    int dtype;
    int ndim;
    size_t dims[3];
    size_t n, nh, nlast, nhl;   // elements, half-spectrum elements, last axis, last axis half
    void* wbar;                 // [nh] complex of dtype
    double* s2;                 // [nh] spectrum (double)
    double s_mean;              // mean of the full spectrum
    std::vector<ConvPlans> plans;
    void* work;                 // fz, fx, fo [batch, nh] complex + a_raw[batch] + partial sums
    size_t work_batch, proj_batch;
    void* fz;
    void* fx;
    void* fo;
    void* fr;                   // [batch, n] real: the mean when b_new may alias b_old (damping)
    // own two-pass FFT engine (1-D, n / 2 = 2^a 5^b up to ~1.3 10^6): half-length complex transforms
    // of the packed signal, spectrum kept in the transform's permuted order (fft_pow2.hip)
    Fft2Plan* fft2;             // null: rocFFT engine
    size_t h;                   // n / 2
    void* wbar_p;               // [h] filter bins in permuted order (bin kfreq(p) at position p)
    double* s2_p;               // [h]
    double wbar_nyq, s2_nyq;    // bin h (real)
    int* partner;               // [h] position holding frequency h - k
    int* leaders;               // [nlead] 128-byte lines of the spectrum that lead a (line, partner line) pair
    size_t nlead;
    bool lines_closed;
    void* wk;                   // [h] unpack twiddle exp(-2 pi i kfreq(p) / n)
    // own engine in 2 / 3 dimensions (and short 1-D signals): FftNdPlan, every axis in its own permuted order; the
    // tables above hold [D][h] filter bins (D = 1, or the d filters of the gradient channel), wk [hl] by last-axis
    // position, plus the last-axis Nyquist bins of every outer position
    FftNdPlan* fftn;
    bool cplx;                  // real=False: complex signals and filter, full spectra [n], planar vectors [2, n]
    size_t hl;                  // n_last / 2
    int ncomp;                  // D
    void* wbar_nq;              // [D][h / hl] complex
    double* s2_nq;              // [h / hl]
    void* fzc;                  // transforms of the constant part of a damped constant message
    void* fxc;
    bool fzc_valid, fxc_valid;
    double* a_raw;
    double* partial;
    int nblk;
};

namespace {

int rocfft_fail(tramp_ctx* ctx, const char* what, rocfft_status st) {
    char msg[64];
    snprintf(msg, sizeof(msg), "rocfft status %d", (int)st);
    return tramp_fail(ctx, TRAMP_EHIP, what, msg);
}

#define TRAMP_FFT_CHECK(ctx, call)                                   \
    do {                                                             \
        rocfft_status st_ = (call);                                  \
        if (st_ != rocfft_status_success) return rocfft_fail(ctx, #call, st_); \
    } while (0)

int make_plan_dims(tramp_ctx* ctx, int ndim, const size_t* dims, rocfft_transform_type type, bool f64, size_t batch,
                   rocfft_plan* plan, rocfft_execution_info* info, void** work) {
    size_t lengths[3];
    for (int d = 0; d < ndim; ++d) lengths[d] = dims[ndim - 1 - d];   // rocFFT: fastest axis first
    TRAMP_FFT_CHECK(ctx, rocfft_plan_create(plan, rocfft_placement_notinplace, type,
                                            f64 ? rocfft_precision_double : rocfft_precision_single,
                                            (size_t)ndim, lengths, batch, nullptr));
    size_t wsz = 0;
    TRAMP_FFT_CHECK(ctx, rocfft_plan_get_work_buffer_size(*plan, &wsz));
    TRAMP_FFT_CHECK(ctx, rocfft_execution_info_create(info));
    *work = nullptr;
    if (wsz) {
        TRAMP_HIP_CHECK(ctx, hipMalloc(work, wsz));
        TRAMP_FFT_CHECK(ctx, rocfft_execution_info_set_work_buffer(*info, *work, wsz));
    }
    TRAMP_FFT_CHECK(ctx, rocfft_execution_info_set_stream(*info, ctx->stream));
    return TRAMP_OK;
}

int make_plan(tramp_ctx* ctx, const tramp_conv* cv, bool forward, bool f64, size_t batch, rocfft_plan* plan,
              rocfft_execution_info* info, void** work) {
    if (cv->cplx)
        return make_plan_dims(ctx, cv->ndim, cv->dims,
                              forward ? rocfft_transform_type_complex_forward : rocfft_transform_type_complex_inverse, f64,
                              batch, plan, info, work);
    return make_plan_dims(ctx, cv->ndim, cv->dims,
                          forward ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse, f64,
                          batch, plan, info, work);
}

int plans_for(tramp_ctx* ctx, tramp_conv* cv, size_t batch, ConvPlans** out) {
    for (auto& p : cv->plans)
        if (p.batch == batch) {
            *out = &p;
            return TRAMP_OK;
        }
    if (ctx->capturing) return tramp_fail(ctx, TRAMP_EINVAL, "tramp_conv", "FFT plan creation inside graph capture");
    ConvPlans p{};
    p.batch = batch;
    const bool f64 = cv->dtype == TRAMP_F64;
    int rc = make_plan(ctx, cv, true, f64, batch, &p.fwd, &p.info_fwd, &p.work_fwd);
    if (rc) return rc;
    rc = make_plan(ctx, cv, false, f64, batch, &p.inv, &p.info_inv, &p.work_inv);
    if (rc) return rc;
    cv->plans.push_back(p);
    *out = &cv->plans.back();
    return TRAMP_OK;
}

int fft_exec(tramp_ctx* ctx, rocfft_plan plan, rocfft_execution_info info, const void* in, void* out) {
    void* ib[1] = {const_cast<void*>(in)};
    void* ob[1] = {out};
    TRAMP_FFT_CHECK(ctx, rocfft_execute(plan, ib, ob, info));
    return TRAMP_OK;
}

int conv_reserve(tramp_ctx* ctx, tramp_conv* cv, size_t batch) {
    if (batch <= cv->work_batch) return TRAMP_OK;
    if (ctx->capturing) return tramp_fail(ctx, TRAMP_EINVAL, "tramp_conv", "workspace growth inside graph capture");
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (cv->work) (void)hipFree(cv->work);
    cv->work = nullptr;
    cv->work_batch = 0;
    const size_t csz = 2 * dtype_size(cv->dtype);
    const size_t per = ((batch * cv->nh * csz + 255) / 256) * 256;
    const size_t scal = ((batch * sizeof(double) + 255) / 256) * 256;
    const size_t part = ((batch * (size_t)cv->nblk * sizeof(double) + 255) / 256) * 256;
    const size_t rper = ((batch * cv->n * (cv->cplx ? 2 : 1) * dtype_size(cv->dtype) + 255) / 256) * 256;
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->work, 5 * per + rper + scal + part));
    char* p = (char*)cv->work;
    cv->fz = p;
    cv->fx = p + per;
    cv->fo = p + 2 * per;
    cv->fzc = p + 3 * per;
    cv->fxc = p + 4 * per;
    cv->fzc_valid = cv->fxc_valid = false;
    cv->fr = p + 5 * per;
    cv->a_raw = (double*)(p + 5 * per + rper);
    cv->partial = (double*)(p + 5 * per + rper + scal);
    cv->work_batch = batch;
    cv->proj_batch = 0;
    return TRAMP_OK;
}

// ---- setup: spectrum and filter transform in the working precision
template <typename T>
__global__ __launch_bounds__(kBlock) void conv_setup_kernel(const double2* __restrict__ w64, size_t nh,
                                                            typename Cx<T>::type* __restrict__ wbar,
                                                            double* __restrict__ s2) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nh) return;
    const double2 w = w64[i];
    typename Cx<T>::type o;
    o.x = (T)w.x;
    o.y = (T)w.y;
    wbar[i] = o;
    s2[i] = w.x * w.x + w.y * w.y;                                     // conv_channel.py:45
}

// Multiplicity of half-spectrum bin j of the last axis in the full spectrum.
__device__ __forceinline__ double herm_weight(size_t j, size_t nlast) {
    return (j == 0 || 2 * j == nlast) ? 1.0 : 2.0;
}

// ---- K9 for the convolution: n_eff = mean(s / (az/ax + s)) over the FULL spectrum
// (conv_channel.py:69-79), two deterministic stages.
__global__ __launch_bounds__(kBlock) void conv_neff_partial_kernel(const double* __restrict__ s2, size_t nh,
                                                                   size_t nhl, size_t nlast,
                                                                   const double* __restrict__ az_d,
                                                                   const double* __restrict__ ax_d,
                                                                   double* __restrict__ partial) {
    __shared__ double smem[kBlock / kWave];
    const size_t b = blockIdx.y;
    const double az = az_d[b], ax = ax_d[b];
    double s = 0.0;
    if (ax != 0.0 && az / ax != 0.0) {
        const double q = az / ax;
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nh; i += (size_t)gridDim.x * kBlock) {
            const double sp = s2[i];
            s += herm_weight(i % nhl, nlast) * (sp / (q + sp));
        }
    }
    s = block_sum(s, smem);
    if (threadIdx.x == 0) partial[b * gridDim.x + blockIdx.x] = s;
}

__global__ __launch_bounds__(kWave) void conv_variance_kernel(const double* __restrict__ partial, int nblk,
                                                              double n, double s_mean, double fwd_div,
                                                              int dir, const double* __restrict__ az_d,
                                                              const double* __restrict__ ax_d, double damping,
                                                              const double* __restrict__ a_old,
                                                              double* __restrict__ v_out, double* __restrict__ a_new_out,
                                                              double* __restrict__ a_new_raw,
                                                              double* __restrict__ n_eff_out, size_t batch) {
    // one wave per instance: lanes fold the partial sums (fixed assignment and tree -> deterministic)
    const size_t b = blockIdx.x;
    const double az = az_d[b], ax = ax_d[b];
    double n_eff;
    if (ax == 0.0) n_eff = 0.0;                                        // :71-73
    else if (az / ax == 0.0) n_eff = 1.0;                              // :74-76
    else {
        double s = 0.0;
        for (int j = threadIdx.x; j < nblk; j += kWave) s += partial[b * nblk + j];
        s = wave_sum(s);
        s = __shfl(s, 0, kWave);
        n_eff = s / n;                                                 // :77-78
    }
    if (threadIdx.x != 0) return;
    double v, a_in;
    if (dir == TRAMP_FWD) {
        // conv_channel.py:117-123; gradient_channel.py:91-97 divides by ax * d (fwd_div = d)
        v = (ax == 0.0) ? s_mean / az : (fwd_div == 1.0 ? n_eff / ax : n_eff / (ax * fwd_div));
        a_in = ax;
    } else {
        v = (1.0 - n_eff) / az;                                        // :111-115
        a_in = az;
    }
    const double an = tmath::a_new_of(v, a_in);
    if (a_new_raw) a_new_raw[b] = an;
    if (a_new_out) a_new_out[b] = (damping != 0.0) ? damping * a_old[b] + (1.0 - damping) * an : an;
    if (v_out) v_out[b] = v;
    if (n_eff_out) n_eff_out[b] = n_eff;
}

// ---- spectral solve.  mode 0: out = conj(wbar) rz_fft / n (forward mean), 1: out = rz_fft / n
// (backward mean), 2: out = conj(wbar) fz / n (plain convolution, sampling).
template <typename T>
__global__ __launch_bounds__(kBlock) void conv_spectral_kernel(int mode, size_t nh, double inv_n,
                                                               const typename Cx<T>::type* __restrict__ wbar,
                                                               const double* __restrict__ s2,
                                                               const double* __restrict__ az_d,
                                                               const double* __restrict__ ax_d,
                                                               const typename Cx<T>::type* __restrict__ fz,
                                                               const typename Cx<T>::type* __restrict__ fx,
                                                               typename Cx<T>::type* __restrict__ fo) {
    using C = typename Cx<T>::type;
    const size_t b = blockIdx.y;
    const double az = (mode == 2) ? 0.0 : az_d[b];
    const double ax = (mode == 2) ? 0.0 : ax_d[b];
    const size_t base = b * nh;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nh; i += (size_t)gridDim.x * kBlock) {
        const C w = wbar[i];
        const C z = fz[base + i];
        C o;
        if (mode == 2) {
            const T sc = (T)inv_n;
            o.x = (w.x * z.x + w.y * z.y) * sc;                        // conj(w) z
            o.y = (w.x * z.y - w.y * z.x) * sc;
        } else {
            const C x = fx[base + i];
            const T res = (T)(inv_n / (az + ax * s2[i]));              // resolvent, :92
            C rz;
            rz.x = (z.x + (w.x * x.x - w.y * x.y)) * res;              // (fz + wbar fx) resolvent, :93
            rz.y = (z.y + (w.x * x.y + w.y * x.x)) * res;
            if (mode == 1) o = rz;
            else {
                o.x = w.x * rz.x + w.y * rz.y;                         // w_fft = conj(wbar), :107
                o.y = w.x * rz.y - w.y * rz.x;
            }
        }
        fo[base + i] = o;
    }
}

// ---- spectral solve on the PERMUTED half-length spectrum of the packed signal (own FFT engine).
// z[j] = x[2j] + i x[2j+1] has the length-h transform Z; with E, O the transforms of the even and
// odd samples, Z_k = E_k + i O_k and the real signal's bins are X_k = E_k + w^k O_k, w = exp(-2 pi i / n):
//     X_k = ((Z_k + conj Z_{h-k}) - i w^k (Z_k - conj Z_{h-k})) / 2,   X_0 = Re Z_0 + Im Z_0,  X_h = Re Z_0 - Im Z_0
// and back Z'_k = ((Y_k + conj Y_{h-k}) + i conj(w^k) (Y_k - conj Y_{h-k})) / 2.  A thread owns the pair
// (k, h-k), k <= h/2, through the partner table; modes as conv_spectral_kernel.
template <typename C> __device__ __forceinline__ C cx_mul(C a, C b) {
    C r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
template <typename C> __device__ __forceinline__ C cx_conj(C a) { a.y = -a.y; return a; }

template <typename T>
__device__ __forceinline__ typename Cx<T>::type spectral_bin(int mode, typename Cx<T>::type xz,
                                                            typename Cx<T>::type xx, typename Cx<T>::type w,
                                                            double s, double az, double ax, double inv) {
    using C = typename Cx<T>::type;
    C o;
    if (mode == 2) {
        o = cx_mul(cx_conj(w), xz);
        o.x *= (T)inv;
        o.y *= (T)inv;
        return o;
    }
    const T res = (T)(inv / (az + ax * s));
    const C wx = cx_mul(w, xx);
    C rz;
    rz.x = (xz.x + wx.x) * res;
    rz.y = (xz.y + wx.y) * res;
    return mode == 1 ? rz : cx_mul(cx_conj(w), rz);
}

// Work is handed out by 128-byte LINES of the permuted spectrum: the frequencies of a line differ only in
// their top bits, so all their partners h - k sit in ONE other line; a group of lanes takes a leading line
// and its partner line whole (full-line loads and stores for both), instead of half lines twice.
template <typename T>
__global__ __launch_bounds__(kBlock) void conv_spectral_perm_kernel(int mode, size_t h, double inv_h,
                                                                    const typename Cx<T>::type* __restrict__ wbar_p,
                                                                    const double* __restrict__ s2_p, double wbar_nyq,
                                                                    double s2_nyq, const int* __restrict__ partner,
                                                                    const int* __restrict__ leaders, size_t nlead,
                                                                    int closed,
                                                                    const typename Cx<T>::type* __restrict__ wk,
                                                                    const double* __restrict__ az_d,
                                                                    const double* __restrict__ ax_d,
                                                                    const typename Cx<T>::type* __restrict__ fz,
                                                                    const typename Cx<T>::type* __restrict__ fx,
                                                                    typename Cx<T>::type* __restrict__ fo, size_t batch,
                                                                    int ipb) {
    using C = typename Cx<T>::type;
    constexpr int LINE = 128 / (int)sizeof(C);
    // blockIdx.y owns `ipb` consecutive instances: the per-bin tables (filter spectrum, |w|^2, twiddle, partner:
    // 44 bytes per bin in f64, as much as the three spectra of ONE instance) are read once per group
    const size_t b_lo = (size_t)blockIdx.y * ipb;
    const size_t b_hi = (b_lo + ipb < batch) ? b_lo + ipb : batch;
    const C zero = {};
    auto unpack = [](C a, C bq, C tw) {           // ((a + conj b) - i tw (a - conj b)) / 2
        const C cb = cx_conj(bq);
        C sum = {a.x + cb.x, a.y + cb.y}, dif = {a.x - cb.x, a.y - cb.y};
        const C td = cx_mul(tw, dif);             // -i td = (td.y, -td.x)
        C r = {T(0.5) * (sum.x + td.y), T(0.5) * (sum.y - td.x)};
        return r;
    };
    auto repack = [](C y, C yq, C tw) {           // ((y + conj yq) + i conj(tw) (y - conj yq)) / 2
        const C cb = cx_conj(yq);
        C sum = {y.x + cb.x, y.y + cb.y}, dif = {y.x - cb.x, y.y - cb.y};
        const C td = cx_mul(cx_conj(tw), dif);    // i td = (-td.y, td.x)
        C r = {T(0.5) * (sum.x - td.y), T(0.5) * (sum.y + td.x)};
        return r;
    };
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < nlead * LINE; idx += (size_t)gridDim.x * kBlock) {
        const size_t p = (size_t)leaders[idx / LINE] * LINE + (idx % LINE);
        const int pp = partner[p];
        // each pair once: inside a self-paired line (or when the lines do not close) by its lower position
        if ((!closed || (size_t)pp / LINE == p / LINE) && (size_t)pp < p) continue;
        if (p == 0) {                                  // bins 0 and h, both real
            const C w0 = wbar_p[0];
            const double s0 = s2_p[0];
            const C wn = {(T)wbar_nyq, T(0)};
            for (size_t b = b_lo; b < b_hi; ++b) {
                const double az = (mode == 2) ? 0.0 : az_d[b], ax = (mode == 2) ? 0.0 : ax_d[b];
                const C zp = fz[b * h], xp = (mode == 2) ? zero : fx[b * h];
                C xz0 = {zp.x + zp.y, T(0)}, xzh = {zp.x - zp.y, T(0)};
                C xx0 = {xp.x + xp.y, T(0)}, xxh = {xp.x - xp.y, T(0)};
                const C y0 = spectral_bin<T>(mode, xz0, xx0, w0, s0, az, ax, inv_h);
                const C yh = spectral_bin<T>(mode, xzh, xxh, wn, s2_nyq, az, ax, inv_h);
                C o = {T(0.5) * (y0.x + yh.x), T(0.5) * (y0.x - yh.x)};
                fo[b * h] = o;
            }
            continue;
        }
        const C w = wk[p];
        const C wq = {-w.x, w.y};                     // w^(h-k) = -conj(w^k)
        const C wbp = wbar_p[p], wbq = wbar_p[pp];
        const double s2p = s2_p[p], s2q = s2_p[pp];
        for (size_t b = b_lo; b < b_hi; ++b) {
            const double az = (mode == 2) ? 0.0 : az_d[b], ax = (mode == 2) ? 0.0 : ax_d[b];
            const size_t base = b * h;
            const C zp = fz[base + p], xp = (mode == 2) ? zero : fx[base + p];
            const C zq = fz[base + pp], xq = (mode == 2) ? zero : fx[base + pp];
            const C xzk = unpack(zp, zq, w), xzq = unpack(zq, zp, wq);
            const C xxk = (mode == 2) ? zero : unpack(xp, xq, w), xxq = (mode == 2) ? zero : unpack(xq, xp, wq);
            const C yk = spectral_bin<T>(mode, xzk, xxk, wbp, s2p, az, ax, inv_h);
            const C yq = spectral_bin<T>(mode, xzq, xxq, wbq, s2q, az, ax, inv_h);
            fo[base + p] = repack(yk, yq, w);
            if ((size_t)pp != p) fo[base + pp] = repack(yq, yk, wq);
        }
    }
}


// ---- message update on the inverse-transformed mean (tramp/channels/base_channel.py:9-17,
// tramp/base.py:245-254), damping fused; r may be updated in place.
template <typename T>
__global__ __launch_bounds__(kBlock) void conv_epilogue_kernel(size_t n, const double* __restrict__ a_in,
                                                               const double* __restrict__ a_raw,
                                                               const T* __restrict__ b_in, double damping,
                                                               const T* b_old, const T* r, T* out, T* r_out) {
    const size_t b = blockIdx.y;
    const T asum = (T)(a_in[b] + a_raw[b]);
    const T d = (T)damping, omd = (T)(1.0 - damping);
    const size_t base = b * n;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const T rv = r[base + i];
        T bn = rv * asum - b_in[base + i];
        if (damping != 0.0) bn = d * b_old[base + i] + omd * bn;
        out[base + i] = bn;
        if (r_out) r_out[base + i] = rv;
    }
}

unsigned grid_for(size_t n) {
    size_t g = (n + (size_t)kBlock * 4 - 1) / ((size_t)kBlock * 4);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (unsigned)g;
}

int conv_scalars(tramp_ctx* ctx, const tramp_conv* cv, double* partial, int dir, size_t batch, const double* az_d,
                 const double* ax_d, double damping, const double* a_old, double* v_d, double* a_new_d,
                 double* a_raw, double* n_eff_d, double fwd_div = 1.0) {
    hipLaunchKernelGGL(conv_neff_partial_kernel, dim3((unsigned)cv->nblk, (unsigned)batch), dim3(kBlock), 0,
                       ctx->stream, cv->s2, cv->nh, cv->nhl, cv->nlast, az_d, ax_d, partial);
    hipLaunchKernelGGL(conv_variance_kernel, dim3((unsigned)batch), dim3(kWave), 0, ctx->stream, partial,
                       cv->nblk, (double)cv->n, cv->s_mean, fwd_div, dir, az_d, ax_d, damping, a_old, v_d, a_new_d,
                       a_raw, n_eff_d, batch);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

void conv_free_fft2(tramp_conv* cv) {
    if (cv->fft2) fft2_plan_destroy(cv->fft2);
    if (cv->wbar_p) (void)hipFree(cv->wbar_p);
    if (cv->s2_p) (void)hipFree(cv->s2_p);
    if (cv->partner) (void)hipFree(cv->partner);
    if (cv->leaders) (void)hipFree(cv->leaders);
    cv->leaders = nullptr;
    if (cv->wk) (void)hipFree(cv->wk);
    if (cv->fftn) fftnd_plan_destroy(cv->fftn);
    if (cv->wbar_nq) (void)hipFree(cv->wbar_nq);
    if (cv->s2_nq) (void)hipFree(cv->s2_nq);
    cv->fftn = nullptr;
    cv->wbar_nq = nullptr;
    cv->s2_nq = nullptr;
    cv->fft2 = nullptr;
    cv->wbar_p = cv->wk = nullptr;
    cv->s2_p = nullptr;
    cv->partner = nullptr;
}

// Which engine transforms a signal of `points` complex points per instance.  Round 5: the OWN engine wherever it has a
// plan (lengths 2^a 5^b in its range; everything else - odd lengths, primes, signals below its smallest tile - is
// rocFFT's), so no vendor library sits on the iteration path of the shapes it covers.  Measured
// (tools/sweep_engine_defaults.sh, profiles/r05/sweep_engine_defaults.txt): ahead where the transforms are bound by
// memory - 256 x 256 and up 1.05-1.27x, a 10^6-point gradient channel 2x - and 10-30 % behind rocFFT's fused small-size
// kernels on small arrays (128 x 128, 32^3, n <= 8192), which are bound by launches; what the switch buys there is the
// absence of rocFFT's run-time kernel compilation (78 s on a fresh box for the first f64 plan) and one code path.
// TRAMP_CONV_FFT=own / =rocfft force one everywhere (the tests run both); TRAMP_CONV_FFT_MIN_POINTS=32768 restores the
// round-3 threshold.
bool own_engine_wanted(size_t points) {
    const char* eng = getenv("TRAMP_CONV_FFT");
    if (eng && strcmp(eng, "rocfft") == 0) return false;
    if (eng && strcmp(eng, "own") == 0) return true;
    static const size_t min_points = getenv("TRAMP_CONV_FFT_MIN_POINTS") ? (size_t)atol(getenv("TRAMP_CONV_FFT_MIN_POINTS")) : 0;
    return points >= min_points;
}

// Tables of the own-FFT engine: the filter bins and the unpack twiddles in the transform's
// permuted order, and the k <-> h-k partner positions.  w64: natural-order half spectrum (device),
// s2h: its squared modulus (host).
int conv_setup_fft2(tramp_ctx* ctx, tramp_conv* cv, const double2* w64, const std::vector<double>& s2h) {
    if (cv->ndim != 1 || cv->n < 2 || (cv->n & 1)) return TRAMP_OK;
    const size_t h = cv->n / 2;
    if (!own_engine_wanted(h)) return TRAMP_OK;
    Fft2Shape shp;
    // 1024 <= h <= 4096: the whole transform is one block pass (one launch, message update fused into it); above: two passes
    const bool blk = h >= 1024 && fft2_block_shape(h, &shp);
    if (!blk && !fft2_shape(h, &shp)) return TRAMP_OK;   // not 2^a 5^b in the engine's range: the N-D plans or rocFFT
    std::vector<double2> wnat(cv->nh);
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(wnat.data(), w64, cv->nh * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<size_t> kf(h);
    std::vector<int> pos(h), partner(h);
    for (size_t p = 0; p < h; ++p) {
        kf[p] = fft2_kfreq(shp, p);
        pos[kf[p]] = (int)p;
    }
    const bool f64 = cv->dtype == TRAMP_F64;
    std::vector<double> s2p(h);
    std::vector<double2> wp64(h), wk64(h);
    for (size_t p = 0; p < h; ++p) {
        const size_t k = kf[p];
        partner[p] = pos[(h - k) % h];
        wp64[p] = wnat[k];
        s2p[p] = s2h[k];
        const double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)cv->n;
        wk64[p] = double2{cos(ang), sin(ang)};
    }
    cv->wbar_nyq = wnat[h].x;
    cv->s2_nyq = s2h[h];
    // leading lines: line l pairs with the line of the partner of its second position (never frequency 0)
    // (closed = every line's partners lie in exactly one line; otherwise every line leads and a pair is
    //  taken by its lower position - correct for any digit order, only with half-used lines)
    const size_t line = 128 / (2 * dtype_size(cv->dtype));
    bool closed = true;
    for (size_t l = 0; l < h / line && closed; ++l) {
        const size_t lp = (size_t)partner[l * line + 1] / line;
        for (size_t e = 0; e < line; ++e)
            if ((size_t)partner[l * line + e] / line != lp && !(l == 0 && e == 0)) closed = false;
    }
    std::vector<int> leaders;
    for (size_t l = 0; l < h / line; ++l)
        if (!closed || l <= (size_t)partner[l * line + 1] / line) leaders.push_back((int)l);
    cv->nlead = leaders.size();
    cv->lines_closed = closed;
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->leaders, leaders.size() * sizeof(int)));
    TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->leaders, leaders.data(), leaders.size() * sizeof(int), hipMemcpyHostToDevice));
    const size_t csz = 2 * dtype_size(cv->dtype);
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->wbar_p, h * csz));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->wk, h * csz));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->s2_p, h * sizeof(double)));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->partner, h * sizeof(int)));
    if (f64) {
        TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->wbar_p, wp64.data(), h * csz, hipMemcpyHostToDevice));
        TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->wk, wk64.data(), h * csz, hipMemcpyHostToDevice));
    } else {
        std::vector<float2> a(h), b(h);
        for (size_t p = 0; p < h; ++p) {
            a[p] = float2{(float)wp64[p].x, (float)wp64[p].y};
            b[p] = float2{(float)wk64[p].x, (float)wk64[p].y};
        }
        TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->wbar_p, a.data(), h * csz, hipMemcpyHostToDevice));
        TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->wk, b.data(), h * csz, hipMemcpyHostToDevice));
    }
    TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->s2_p, s2p.data(), h * sizeof(double), hipMemcpyHostToDevice));
    TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->partner, partner.data(), h * sizeof(int), hipMemcpyHostToDevice));
    int rc = fft2_plan_create(ctx, cv->dtype, h, &cv->fft2, blk);
    if (rc) return rc;
    cv->h = h;
    return TRAMP_OK;
}

// ---- spectral solve on the permuted packed spectrum in any number of dimensions, D filter components.
// The packed image z[..][j] = x[..][2j] + i x[..][2j+1] has the complex transform Z = E + i O (E, O: transforms of the even
// and odd columns, period hl in the last frequency), X[k] = E[k] + w^kl O[k].  With q = (-k_outer, hl - kl):
// conj Z[q] = E[k] - i O[k], so the pair (position p of k, partner position of q) untangles as in one dimension.  In the
// column kl = 0 the partner is (-k_outer, 0) and the pair carries FOUR bins: (k_outer, 0), (k_outer, hl) and the two of
// -k_outer (w = 1 and w = -1); the Nyquist bins of the filters sit in their own small tables.
//   rz = (fz + sum_c t_c fx_c) / (az + ax s);  mode 0: out_c = conj(t_c) rz, 1: out = rz, 2: out_c = conj(t_c) fz
template <typename T, int D>
__device__ __forceinline__ void spectral_bin_nd(int mode, typename Cx<T>::type xz, const typename Cx<T>::type* xx,
                                                const typename Cx<T>::type* t, double s, double az, double ax, double inv,
                                                typename Cx<T>::type* out) {
    using C = typename Cx<T>::type;
    C rz = xz;
    T res = (T)inv;
    if (mode != 2) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const C wx = cx_mul(t[c], xx[c]);
            rz.x += wx.x;
            rz.y += wx.y;
        }
        res = (T)(inv / (az + ax * s));
    }
    rz.x *= res;
    rz.y *= res;
    if (mode == 1) {
        out[0] = rz;
        return;
    }
#pragma unroll
    for (int c = 0; c < D; ++c) out[c] = cx_mul(cx_conj(t[c]), rz);
}

template <typename T, int D>
__global__ __launch_bounds__(kBlock) void spectral_perm_nd_kernel(
    int mode, size_t h, size_t hl, double inv_h, const typename Cx<T>::type* __restrict__ wbar_p,
    const typename Cx<T>::type* __restrict__ wbar_nq, const double* __restrict__ s2_p, const double* __restrict__ s2_nq,
    const int* __restrict__ partner, const typename Cx<T>::type* __restrict__ wk, const double* __restrict__ az_d,
    const double* __restrict__ ax_d, const typename Cx<T>::type* __restrict__ fz,
    const typename Cx<T>::type* __restrict__ fx, typename Cx<T>::type* __restrict__ fo, size_t batch, int ipb) {
    using C = typename Cx<T>::type;
    const size_t b_lo = (size_t)blockIdx.y * ipb;
    const size_t b_hi = (b_lo + ipb < batch) ? b_lo + ipb : batch;
    const size_t nrow = h / hl;
    const int nout = (mode == 1) ? 1 : D;
    const C zero = {};
    auto unpack = [](C a, C bq, C tw) {           // ((a + conj b) - i tw (a - conj b)) / 2
        const C cb = cx_conj(bq);
        C sum = {a.x + cb.x, a.y + cb.y}, dif = {a.x - cb.x, a.y - cb.y};
        const C td = cx_mul(tw, dif);
        C r = {T(0.5) * (sum.x + td.y), T(0.5) * (sum.y - td.x)};
        return r;
    };
    auto repack = [](C y, C yq, C tw) {           // ((y + conj yq) + i conj(tw) (y - conj yq)) / 2
        const C cb = cx_conj(yq);
        C sum = {y.x + cb.x, y.y + cb.y}, dif = {y.x - cb.x, y.y - cb.y};
        const C td = cx_mul(cx_conj(tw), dif);
        C r = {T(0.5) * (sum.x - td.y), T(0.5) * (sum.y + td.x)};
        return r;
    };
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < h; p += (size_t)gridDim.x * kBlock) {
        const size_t pp = (size_t)partner[p];
        if (pp < p) continue;                       // each pair once, by its lower position
        const bool col0 = (p % hl) == 0;
        C tp[D], tq[D], tnp[D], tnq[D];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            tp[c] = wbar_p[(size_t)c * h + p];
            tq[c] = wbar_p[(size_t)c * h + pp];
            tnp[c] = tnq[c] = zero;
        }
        const double sp = s2_p[p], sq = s2_p[pp];
        double snp = 0.0, snq = 0.0;
        C w = {T(1), T(0)};
        if (col0) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                tnp[c] = wbar_nq[(size_t)c * nrow + p / hl];
                tnq[c] = wbar_nq[(size_t)c * nrow + pp / hl];
            }
            snp = s2_nq[p / hl];
            snq = s2_nq[pp / hl];
        } else {
            w = wk[p % hl];
        }
        const C wq = {-w.x, w.y};                  // w^(hl - kl) = -conj(w^kl); in column 0: the Nyquist twiddle -1
        for (size_t b = b_lo; b < b_hi; ++b) {
            const double az = (mode == 2) ? 0.0 : az_d[b], ax = (mode == 2) ? 0.0 : ax_d[b];
            const C zp = fz[b * h + p], zq = fz[b * h + pp];
            C xp[D], xq[D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                xp[c] = (mode == 2) ? zero : fx[(b * D + c) * h + p];
                xq[c] = (mode == 2) ? zero : fx[(b * D + c) * h + pp];
            }
            C xxk[D], xxq[D], yk[D], yq[D];
            if (!col0) {
                const C xzk = unpack(zp, zq, w), xzq = unpack(zq, zp, wq);
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    xxk[c] = unpack(xp[c], xq[c], w);
                    xxq[c] = unpack(xq[c], xp[c], wq);
                }
                spectral_bin_nd<T, D>(mode, xzk, xxk, tp, sp, az, ax, inv_h, yk);
                spectral_bin_nd<T, D>(mode, xzq, xxq, tq, sq, az, ax, inv_h, yq);
                for (int c = 0; c < nout; ++c) {
                    fo[(b * nout + c) * h + p] = repack(yk[c], yq[c], w);
                    if (pp != p) fo[(b * nout + c) * h + pp] = repack(yq[c], yk[c], wq);
                }
            } else {
                // bins (k, 0) and (k, hl) of this row [w = 1, -1], then the same two of the partner row
                C ya[D], yb[D];
                for (int side = 0; side < 2; ++side) {
                    const C a = side ? zq : zp, bq = side ? zp : zq;
                    const C xz0 = unpack(a, bq, w), xzn = unpack(a, bq, wq);
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        xxk[c] = unpack(side ? xq[c] : xp[c], side ? xp[c] : xq[c], w);
                        xxq[c] = unpack(side ? xq[c] : xp[c], side ? xp[c] : xq[c], wq);
                    }
                    spectral_bin_nd<T, D>(mode, xz0, xxk, side ? tq : tp, side ? sq : sp, az, ax, inv_h, ya);
                    spectral_bin_nd<T, D>(mode, xzn, xxq, side ? tnq : tnp, side ? snq : snp, az, ax, inv_h, yb);
                    for (int c = 0; c < nout; ++c) {
                        // Z' = E' + i O',  E' = (Y0 + Yn) / 2,  O' = (Y0 - Yn) / 2
                        const C e = {T(0.5) * (ya[c].x + yb[c].x), T(0.5) * (ya[c].y + yb[c].y)};
                        const C o = {T(0.5) * (ya[c].x - yb[c].x), T(0.5) * (ya[c].y - yb[c].y)};
                        const C zo = {e.x - o.y, e.y + o.x};
                        fo[(b * nout + c) * h + (side ? pp : p)] = zo;
                    }
                    if (pp == p) break;
                }
            }
        }
    }
}

// Tables of the N-D own engine.  D components; w64: natural-order half spectra [D][nh] (device, f64), s2h: the (summed)
// squared modulus [nh] on the host.  conv_1d: the one-dimensional convolution channel keeps its two-pass plan with
// the fused epilogue when that one applies.
int conv_setup_fftn(tramp_ctx* ctx, tramp_conv* cv, int D, const double2* w64, const std::vector<double>& s2h) {
    if (!own_engine_wanted(cv->n / 2)) return TRAMP_OK;
    if (cv->fft2 || !fftnd_supported(cv->dtype, cv->ndim, cv->dims)) return TRAMP_OK;
    if (cv->n / 2 > 0x7fffffffull) return TRAMP_OK;          // partner table is int
    int rc = fftnd_plan_create(ctx, cv->dtype, cv->ndim, cv->dims, &cv->fftn);
    if (rc) return rc;
    const FftNdPlan* pl = cv->fftn;
    const size_t h = pl->h, hl = cv->nlast / 2, nrow = h / hl, nh = cv->nh, nhl = cv->nhl;
    std::vector<double2> wnat((size_t)D * nh);
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(wnat.data(), w64, wnat.size() * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // per axis: frequency of each position and position of each frequency
    std::vector<size_t> kf[3], pos[3];
    for (int a = 0; a < cv->ndim; ++a) {
        const size_t len = pl->dims[a];
        kf[a].resize(len);
        pos[a].resize(len);
        for (size_t i = 0; i < len; ++i) {
            kf[a][i] = fftnd_kfreq(pl, a, i);
            pos[a][kf[a][i]] = i;
        }
    }
    const int nd = cv->ndim;
    std::vector<int> partner(h);
    std::vector<double> s2p(h), s2n(nrow);
    std::vector<double2> wp((size_t)D * h), wn((size_t)D * nrow), wk64(hl);
    for (size_t il = 0; il < hl; ++il) {
        const double ang = -2.0 * 3.14159265358979323846 * (double)kf[nd - 1][il] / (double)cv->nlast;
        wk64[il] = double2{cos(ang), sin(ang)};
    }
    for (size_t row = 0; row < nrow; ++row) {
        // outer positions of this row, their frequencies, the natural row index and the partner row
        size_t rem = row, nat_row = 0, prow = 0, stride_nat = 1, stride_pos = 1;
        for (int a = nd - 2; a >= 0; --a) {
            const size_t len = pl->dims[a], i = rem % len, k = kf[a][i];
            rem /= len;
            nat_row += k * stride_nat;
            prow += pos[a][(len - k) % len] * stride_pos;
            stride_nat *= len;
            stride_pos *= len;
        }
        for (int c = 0; c < D; ++c) wn[(size_t)c * nrow + row] = wnat[(size_t)c * nh + nat_row * nhl + hl];
        s2n[row] = s2h[nat_row * nhl + hl];
        for (size_t il = 0; il < hl; ++il) {
            const size_t kl = kf[nd - 1][il], p = row * hl + il;
            partner[p] = (int)(prow * hl + pos[nd - 1][(hl - kl) % hl]);
            for (int c = 0; c < D; ++c) wp[(size_t)c * h + p] = wnat[(size_t)c * nh + nat_row * nhl + kl];
            s2p[p] = s2h[nat_row * nhl + kl];
        }
    }
    const bool f64 = cv->dtype == TRAMP_F64;
    const size_t csz = 2 * dtype_size(cv->dtype);
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->wbar_p, (size_t)D * h * csz));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->wbar_nq, (size_t)D * nrow * csz));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->wk, hl * csz));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->s2_p, h * sizeof(double)));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->s2_nq, nrow * sizeof(double)));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&cv->partner, h * sizeof(int)));
    auto put = [&](void* dst, const std::vector<double2>& src) -> hipError_t {
        if (f64) return hipMemcpy(dst, src.data(), src.size() * sizeof(double2), hipMemcpyHostToDevice);
        std::vector<float2> t(src.size());
        for (size_t i = 0; i < src.size(); ++i) t[i] = float2{(float)src[i].x, (float)src[i].y};
        return hipMemcpy(dst, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice);
    };
    TRAMP_HIP_CHECK(ctx, put(cv->wbar_p, wp));
    TRAMP_HIP_CHECK(ctx, put(cv->wbar_nq, wn));
    TRAMP_HIP_CHECK(ctx, put(cv->wk, wk64));
    TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->s2_p, s2p.data(), h * sizeof(double), hipMemcpyHostToDevice));
    TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->s2_nq, s2n.data(), nrow * sizeof(double), hipMemcpyHostToDevice));
    TRAMP_HIP_CHECK(ctx, hipMemcpy(cv->partner, partner.data(), h * sizeof(int), hipMemcpyHostToDevice));
    cv->h = h;
    cv->hl = hl;
    cv->ncomp = D;
    return TRAMP_OK;
}

template <typename T, int D>
void launch_spectral_nd_t(tramp_ctx* ctx, const tramp_conv* cv, int mode, size_t batch, const double* az_d,
                          const double* ax_d, const void* fz, const void* fx, void* fo) {
    using C = typename Cx<T>::type;
    int ipb = 8;
    while (ipb > 1 && (size_t)grid_for(cv->h) * ((batch + ipb - 1) / ipb) < 2048) ipb >>= 1;
    dim3 g(grid_for(cv->h), (unsigned)((batch + ipb - 1) / ipb));
    hipLaunchKernelGGL((spectral_perm_nd_kernel<T, D>), g, dim3(kBlock), 0, ctx->stream, mode, cv->h, cv->hl, 1.0 / cv->h,
                       (const C*)cv->wbar_p, (const C*)cv->wbar_nq, cv->s2_p, cv->s2_nq, cv->partner, (const C*)cv->wk, az_d,
                       ax_d, (const C*)fz, (const C*)fx, (C*)fo, batch, ipb);
}

template <typename T>
void launch_spectral_nd(tramp_ctx* ctx, const tramp_conv* cv, int mode, size_t batch, const double* az_d,
                        const double* ax_d, const void* fz, const void* fx, void* fo) {
    if (cv->ncomp == 1) launch_spectral_nd_t<T, 1>(ctx, cv, mode, batch, az_d, ax_d, fz, fx, fo);
    else if (cv->ncomp == 2) launch_spectral_nd_t<T, 2>(ctx, cv, mode, batch, az_d, ax_d, fz, fx, fo);
    else launch_spectral_nd_t<T, 3>(ctx, cv, mode, batch, az_d, ax_d, fz, fx, fo);
}

// forward transform of [batch, n] reals into the engine's spectrum buffer
int conv_forward(tramp_ctx* ctx, tramp_conv* cv, ConvPlans* p, size_t batch, const void* in, void* spec) {
    if (cv->fft2) return fft2_forward(ctx, cv->fft2, batch, in, spec);
    if (cv->fftn) return fftnd_forward(ctx, cv->fftn, batch, in, spec);
    return fft_exec(ctx, p->fwd, p->info_fwd, in, spec);
}

int conv_inverse(tramp_ctx* ctx, tramp_conv* cv, ConvPlans* p, size_t batch, void* spec, void* out) {
    if (cv->fft2) return fft2_inverse(ctx, cv->fft2, batch, spec, out);
    if (cv->fftn) return fftnd_inverse(ctx, cv->fftn, batch, spec, out);
    return fft_exec(ctx, p->inv, p->info_inv, spec, out);
}

template <typename T>
void launch_spectral(tramp_ctx* ctx, tramp_conv* cv, int mode, size_t batch, const double* az_d, const double* ax_d) {
    using C = typename Cx<T>::type;
    if (cv->fft2) {
        // instances per block: as many as keep >= ~2000 blocks in flight, at most 8
        static const int ipb_env = getenv("TRAMP_CONV_IPB") ? atoi(getenv("TRAMP_CONV_IPB")) : 0;
        int ipb = 8;
        while (ipb > 1 && (size_t)grid_for(cv->h / 2) * ((batch + ipb - 1) / ipb) < 2048) ipb >>= 1;
        if (ipb_env > 0) ipb = ipb_env;
        dim3 g(grid_for(cv->h / 2), (unsigned)((batch + ipb - 1) / ipb));
        hipLaunchKernelGGL((conv_spectral_perm_kernel<T>), g, dim3(kBlock), 0, ctx->stream, mode, cv->h, 1.0 / cv->h,
                           (const C*)cv->wbar_p, cv->s2_p, cv->wbar_nyq, cv->s2_nyq, cv->partner, cv->leaders, cv->nlead,
                           cv->lines_closed ? 1 : 0, (const C*)cv->wk, az_d, ax_d, (const C*)cv->fz, (const C*)cv->fx, (C*)cv->fo,
                           batch, ipb);
    } else if (cv->fftn) {
        launch_spectral_nd<T>(ctx, cv, mode, batch, az_d, ax_d, cv->fz, cv->fx, cv->fo);
    } else {
        dim3 g(grid_for(cv->nh), (unsigned)batch);
        hipLaunchKernelGGL((conv_spectral_kernel<T>), g, dim3(kBlock), 0, ctx->stream, mode, cv->nh, 1.0 / cv->n,
                           (const C*)cv->wbar, cv->s2, az_d, ax_d, (const C*)cv->fz, (const C*)cv->fx, (C*)cv->fo);
    }
}

}  // namespace

// complex channel (real=False), defined at the end of this file.  mode 0 / 1: forward / backward message, 2: x = w * z
int conv_complex_run(tramp_ctx* ctx, tramp_conv* cv, int mode, size_t batch, const double* az_d, const void* bz_d,
                     const double* ax_d, const void* bx_d, double damping, const double* a_old_d, const void* b_old_d,
                     double* a_new_d, void* b_new_d, void* r_d, double* v_d, int reuse);

extern "C" int tramp_conv_create(tramp_ctx* ctx, int dtype, int ndim, const size_t* dims, const double* filter_d,
                                 tramp_conv** conv) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, conv && dims && filter_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F64 || dtype == TRAMP_F32);
    TRAMP_REQUIRE(ctx, ndim >= 1 && ndim <= 3);
    std::call_once(g_rocfft_once, [] { (void)rocfft_setup(); });
    tramp_conv* cv = new tramp_conv();
    cv->dtype = dtype;
    cv->ndim = ndim;
    cv->n = 1;
    for (int d = 0; d < ndim; ++d) {
        if (dims[d] == 0) {
            delete cv;
            return tramp_fail(ctx, TRAMP_EINVAL, "tramp_conv_create", "zero-length axis");
        }
        cv->dims[d] = dims[d];
        cv->n *= dims[d];
    }
    cv->nlast = dims[ndim - 1];
    cv->nhl = cv->nlast / 2 + 1;
    cv->nh = cv->n / cv->nlast * cv->nhl;
    cv->nblk = (int)((cv->nh + 4095) / 4096 > 256 ? 256 : (cv->nh + 4095) / 4096);
    cv->work = nullptr;
    cv->work_batch = cv->proj_batch = 0;
    cv->wbar = nullptr;
    cv->s2 = nullptr;
    cv->fft2 = nullptr;
    cv->fftn = nullptr;
    cv->cplx = false;
    cv->hl = 0;
    cv->ncomp = 1;
    cv->wbar_nq = nullptr;
    cv->s2_nq = nullptr;
    cv->h = 0;
    cv->wbar_p = nullptr;
    cv->s2_p = nullptr;
    cv->partner = nullptr;
    cv->leaders = nullptr;
    cv->nlead = 0;
    cv->wk = nullptr;
    // filter transform in f64 regardless of the working precision (the scalars are f64)
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void* pwork = nullptr;
    double2* w64 = nullptr;
    auto cleanup = [&](bool failed) {
        if (w64) (void)hipFree(w64);
        if (pwork) (void)hipFree(pwork);
        if (info) (void)rocfft_execution_info_destroy(info);
        if (plan) (void)rocfft_plan_destroy(plan);
        if (failed) {
            if (cv->wbar) (void)hipFree(cv->wbar);
            if (cv->s2) (void)hipFree(cv->s2);
            conv_free_fft2(cv);
            delete cv;
        }
    };
    int rc = make_plan(ctx, cv, true, true, 1, &plan, &info, &pwork);
    if (rc) {
        cleanup(true);
        return rc;
    }
    const size_t csz = 2 * dtype_size(dtype);
    hipError_t e = hipMalloc(&w64, cv->nh * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc(&cv->wbar, cv->nh * csz);
    if (e == hipSuccess) e = hipMalloc(&cv->s2, cv->nh * sizeof(double));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        cleanup(true);
        return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_conv_create", hipGetErrorString(e));
    }
    rc = fft_exec(ctx, plan, info, filter_d, w64);
    if (rc) {
        cleanup(true);
        return rc;
    }
    const unsigned g = (unsigned)((cv->nh + kBlock - 1) / kBlock);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((conv_setup_kernel<double>), dim3(g), dim3(kBlock), 0, ctx->stream, w64, cv->nh,
                           (double2*)cv->wbar, cv->s2);
    else
        hipLaunchKernelGGL((conv_setup_kernel<float>), dim3(g), dim3(kBlock), 0, ctx->stream, w64, cv->nh,
                           (float2*)cv->wbar, cv->s2);
    // mean of the full spectrum (second_moment, ax = 0 branch): host fold, setup only
    std::vector<double> s2h(cv->nh);
    e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(s2h.data(), cv->s2, cv->nh * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        cleanup(true);
        return tramp_fail(ctx, TRAMP_EHIP, "tramp_conv_create", hipGetErrorString(e));
    }
    long double acc = 0.0L;
    for (size_t i = 0; i < cv->nh; ++i) {
        const size_t j = i % cv->nhl;
        acc += ((j == 0 || 2 * j == cv->nlast) ? 1.0L : 2.0L) * (long double)s2h[i];
    }
    cv->s_mean = (double)(acc / (long double)cv->n);
    // own FFT engine for 1-D power-of-two sizes (TRAMP_CONV_FFT=rocfft keeps the library transforms)
    rc = conv_setup_fft2(ctx, cv, w64, s2h);
    if (!rc) rc = conv_setup_fftn(ctx, cv, 1, w64, s2h);     // 2-D / 3-D, and short 1-D signals
    if (rc) {
        cleanup(true);
        return rc;
    }
    cleanup(false);
    *conv = cv;
    return TRAMP_OK;
}

extern "C" int tramp_conv_destroy(tramp_ctx* ctx, tramp_conv* cv) {
    if (!cv) return TRAMP_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (auto& p : cv->plans) {
        (void)rocfft_execution_info_destroy(p.info_fwd);
        (void)rocfft_execution_info_destroy(p.info_inv);
        (void)rocfft_plan_destroy(p.fwd);
        (void)rocfft_plan_destroy(p.inv);
        if (p.work_fwd) (void)hipFree(p.work_fwd);
        if (p.work_inv) (void)hipFree(p.work_inv);
    }
    if (cv->work) (void)hipFree(cv->work);
    if (cv->wbar) (void)hipFree(cv->wbar);
    if (cv->s2) (void)hipFree(cv->s2);
    conv_free_fft2(cv);
    delete cv;
    return TRAMP_OK;
}

extern "C" int tramp_conv_spectrum(tramp_ctx* ctx, const tramp_conv* cv, double* s2_d, size_t* count,
                                   double* mean) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, cv);
    if (count) *count = cv->nh;
    if (mean) *mean = cv->s_mean;
    if (s2_d)
        TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(s2_d, cv->s2, cv->nh * sizeof(double), hipMemcpyDeviceToDevice,
                                            ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_conv_variance(tramp_ctx* ctx, tramp_conv* cv, int dir, size_t batch, const double* az_d,
                                   const double* ax_d, double* v_d, double* a_new_d, double* n_eff_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, cv && az_d && ax_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    if (!batch) return TRAMP_OK;
    int rc = tramp_ws_reserve(ctx, batch * (size_t)cv->nblk * sizeof(double));
    if (rc) return rc;
    return conv_scalars(ctx, cv, ctx->ws, dir, batch, az_d, ax_d, 0.0, nullptr, v_d, a_new_d, nullptr, n_eff_d);
}

extern "C" int tramp_conv_apply(tramp_ctx* ctx, tramp_conv* cv, size_t batch, const void* z_d, void* x_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, cv && z_d && x_d);
    if (!batch) return TRAMP_OK;
    if (cv->cplx) return conv_complex_run(ctx, cv, 2, batch, nullptr, z_d, nullptr, nullptr, 0.0, nullptr, nullptr, nullptr, x_d,
                                          nullptr, nullptr, 0);
    int rc = conv_reserve(ctx, cv, batch);
    if (rc) return rc;
    ConvPlans* p = nullptr;
    if (!cv->fft2 && !cv->fftn) {
        rc = plans_for(ctx, cv, batch, &p);
        if (rc) return rc;
    }
    cv->proj_batch = 0;                                               // fz is overwritten below
    rc = conv_forward(ctx, cv, p, batch, z_d, cv->fz);
    if (rc) return rc;
    if (cv->dtype == TRAMP_F64) launch_spectral<double>(ctx, cv, 2, batch, nullptr, nullptr);
    else launch_spectral<float>(ctx, cv, 2, batch, nullptr, nullptr);
    TRAMP_LAUNCH_CHECK(ctx);
    return conv_inverse(ctx, cv, p, batch, cv->fo, x_d);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void conv_axpby_kernel(size_t count, T d, T omd, T* __restrict__ p,
                                                            const T* __restrict__ pc) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        p[i] = d * p[i] + omd * pc[i];
}

extern "C" int tramp_conv_project_affine(tramp_ctx* ctx, tramp_conv* cv, int which, size_t batch, const void* c_d,
                                         double damping, int c_changed) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, cv && c_d);
    TRAMP_REQUIRE(ctx, which == TRAMP_REUSE_T || which == TRAMP_REUSE_U);
    if (!batch) return TRAMP_OK;
    if (cv->cplx) return tramp_fail(ctx, TRAMP_ENOTSUP, "tramp_conv_project_affine", "complex channel");
    if (!cv->work || cv->proj_batch != batch || batch > cv->work_batch)
        return tramp_fail(ctx, TRAMP_EINVAL, "tramp_conv_project_affine",
                          "the handle holds no transform of the message being updated");
    const bool is_t = which == TRAMP_REUSE_T;
    bool& valid = is_t ? cv->fzc_valid : cv->fxc_valid;
    void* pc = is_t ? cv->fzc : cv->fxc;
    void* p = is_t ? cv->fz : cv->fx;
    const size_t esz = dtype_size(cv->dtype);
    if (c_changed || !valid) {
        ConvPlans* pl = nullptr;
        int rc = (cv->fft2 || cv->fftn) ? TRAMP_OK : plans_for(ctx, cv, batch, &pl);
        if (rc) return rc;
        ProfScope ps(ctx, PROF_FFT, (double)batch * (cv->n * esz + cv->nh * 2 * esz), 0);
        rc = conv_forward(ctx, cv, pl, batch, c_d, pc);
        if (rc) return rc;
        valid = true;
    }
    const size_t count = batch * ((cv->fft2 || cv->fftn) ? cv->h : cv->nh) * 2;      // real and imaginary parts alike
    ProfScope ps(ctx, PROF_LIN_SCALARS, 3.0 * count * esz, 0);
    if (cv->dtype == TRAMP_F64)
        hipLaunchKernelGGL((conv_axpby_kernel<double>), dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream, count,
                           damping, 1.0 - damping, (double*)p, (const double*)pc);
    else
        hipLaunchKernelGGL((conv_axpby_kernel<float>), dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream, count,
                           (float)damping, (float)(1.0 - damping), (float*)p, (const float*)pc);
    TRAMP_LAUNCH_CHECK(ctx);
    return TRAMP_OK;
}

extern "C" int tramp_conv_message(tramp_ctx* ctx, tramp_conv* cv, int dir, size_t batch, const double* az_d,
                                  const void* bz_d, const double* ax_d, const void* bx_d, double damping,
                                  const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d,
                                  void* r_d, double* v_d, int reuse) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, cv && az_d && bz_d && ax_d && bx_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!batch) return TRAMP_OK;
    if (cv->cplx)
        return conv_complex_run(ctx, cv, dir == TRAMP_FWD ? 0 : 1, batch, az_d, bz_d, ax_d, bx_d, damping, a_old_d, b_old_d,
                                a_new_d, b_new_d, r_d, v_d, reuse);
    const void* work_before = cv->work;
    int rc = conv_reserve(ctx, cv, batch);
    if (rc) return rc;
    if (cv->work != work_before || cv->proj_batch != batch) reuse = 0;
    cv->proj_batch = batch;
    ConvPlans* p = nullptr;
    if (!cv->fft2 && !cv->fftn) {
        rc = plans_for(ctx, cv, batch, &p);
        if (rc) return rc;
    }
    const size_t n = cv->n, nh = cv->nh;
    const size_t esz = dtype_size(cv->dtype);
    // 1. scalars
    {
        ProfScope ps(ctx, PROF_LIN_SCALARS, 0, 0);
        rc = conv_scalars(ctx, cv, cv->partial, dir, batch, az_d, ax_d, damping, a_old_d, v_d, a_new_d, cv->a_raw,
                          nullptr);
        if (rc) return rc;
    }
    // 2. transforms of the incoming messages (shared by the two directions of one iteration)
    if (!(reuse & TRAMP_REUSE_T)) {
        ProfScope ps(ctx, PROF_FFT, (double)batch * (n * esz + nh * 2 * esz), 0);
        rc = conv_forward(ctx, cv, p, batch, bz_d, cv->fz);
        if (rc) return rc;
    }
    if (!(reuse & TRAMP_REUSE_U)) {
        ProfScope ps(ctx, PROF_FFT, (double)batch * (n * esz + nh * 2 * esz), 0);
        rc = conv_forward(ctx, cv, p, batch, bx_d, cv->fx);
        if (rc) return rc;
    }
    // 3. spectral solve
    {
        ProfScope ps(ctx, PROF_LIN_SCALARS, (double)batch * nh * 6 * esz + nh * (2 * esz + 8.0), 0);
        const int mode = (dir == TRAMP_FWD) ? 0 : 1;
        if (cv->dtype == TRAMP_F64) launch_spectral<double>(ctx, cv, mode, batch, az_d, ax_d);
        else launch_spectral<float>(ctx, cv, mode, batch, az_d, ax_d);
        TRAMP_LAUNCH_CHECK(ctx);
    }
    const double* a_in = (dir == TRAMP_FWD) ? ax_d : az_d;
    const void* b_in = (dir == TRAMP_FWD) ? bx_d : bz_d;
    if (cv->fft2) {
        // 4+5. own engine: the inverse transform's last pass applies the message update as it stores
        ProfScope ps(ctx, PROF_FFT, (double)batch * (n * esz * (damping != 0.0 ? 3 : 2) + nh * 2 * esz), 0);
        Fft2Epi epi{a_in, cv->a_raw, b_in, damping, b_old_d, b_new_d, r_d, PriorHintDev{}, 0};
        rc = fft2_inverse(ctx, cv->fft2, batch, cv->fo, nullptr, &epi);
        if (rc) return rc;
        return TRAMP_OK;
    }
    if (cv->fftn) {
        // the same on the N-D plans: the last (row) pass of the inverse stores the message
        ProfScope ps(ctx, PROF_FFT, (double)batch * (n * esz * (damping != 0.0 ? 3 : 2) + nh * 2 * esz), 0);
        Fft2Epi epi{a_in, cv->a_raw, b_in, damping, b_old_d, b_new_d, r_d, PriorHintDev{}, 0};
        return fftnd_inverse(ctx, cv->fftn, batch, cv->fo, nullptr, &epi);
    }
    // 4. back to the signal domain; the mean lands in r_d when asked for, else in b_new_d
    //    (scratch under damping: b_old usually IS b_new and must survive the transform)
    void* mean = r_d ? r_d : (damping != 0.0 ? cv->fr : b_new_d);
    {
        ProfScope ps(ctx, PROF_FFT, (double)batch * (n * esz + nh * 2 * esz), 0);
        rc = conv_inverse(ctx, cv, p, batch, cv->fo, mean);
        if (rc) return rc;
    }
    // 5. message update (in place on the mean)
    {
        ProfScope ps(ctx, PROF_EW, (double)batch * n * esz * (damping != 0.0 ? 4 : 3), 0);
        dim3 g(grid_for(n), (unsigned)batch);
        if (cv->dtype == TRAMP_F64)
            hipLaunchKernelGGL((conv_epilogue_kernel<double>), g, dim3(kBlock), 0, ctx->stream, n, a_in, cv->a_raw,
                               (const double*)b_in, damping, (const double*)b_old_d, (const double*)mean,
                               (double*)b_new_d, (double*)nullptr);
        else
            hipLaunchKernelGGL((conv_epilogue_kernel<float>), g, dim3(kBlock), 0, ctx->stream, n, a_in, cv->a_raw,
                               (const float*)b_in, damping, (const float*)b_old_d, (const float*)mean,
                               (float*)b_new_d, (float*)nullptr);
        TRAMP_LAUNCH_CHECK(ctx);
    }
    return TRAMP_OK;
}

// ---------------------------------------------------------------- gradient channel x = grad z
// GradientChannel (tramp/channels/linear/gradient_channel.py:10-118, real=True): d = ndim circular
// forward-difference filters f_i, x = (x_1 .. x_d) with x_i = w_i * z, fft(w_i) = conj(fft(f_i)).  It is
// diagonal in the Fourier basis like the convolution channel, with the resolvent built on the SUMMED
// spectrum s = sum_i |fft(f_i)|^2:
//     rz_fft = (fft(bz) + sum_i fft(f_i) fft(bx_i)) / (az + ax s),   rx_i = ifft(conj(fft(f_i)) rz_fft),
//     n_eff = mean(s / (az/ax + s)),  vz = (1 - n_eff) / az,  vx = n_eff / (ax d).
// Vectors: z is [batch, n], x is [batch, d, n] (the reference's (d,) + shape, flattened).  The handle embeds
// a tramp_conv for the shape / half-spectrum geometry, the summed spectrum, the n_eff kernels and the
// rocFFT plan cache (batch and batch * d transforms of the same shape).
struct tramp_grad {
    tramp_conv base;
    int d;
    void* wbar_all;             // [d][nh] complex of dtype: fft(f_i)
    void* gwork;                // fz [B,nh], fx [B,d,nh], fo [B,d,nh], fr [B,d,n], a_raw [B], partial
    size_t gwork_batch;
    size_t gproj_batch;         // batch of the transforms currently held in gfz / gfx (TRAMP_REUSE_T / _U)
    void* gfz;
    void* gfx;
    void* gfo;
    void* gfr;
    double* ga_raw;
    double* gpartial;
};

namespace {

template <typename T>
__global__ __launch_bounds__(kBlock) void grad_setup_kernel(const double2* __restrict__ w64, int d, size_t nh,
                                                            typename Cx<T>::type* __restrict__ wbar,
                                                            double* __restrict__ s2) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nh) return;
    double s = 0.0;
    for (int c = 0; c < d; ++c) {
        const double2 w = w64[(size_t)c * nh + i];
        typename Cx<T>::type o;
        o.x = (T)w.x;
        o.y = (T)w.y;
        wbar[(size_t)c * nh + i] = o;
        s += w.x * w.x + w.y * w.y;                                    // gradient_channel.py:23 (sum over axis 0)
    }
    s2[i] = s;
}

// mode 0: fo[b,c,:] = conj(wbar_c) rz_fft / n (forward mean), 1: fo[b,:] = rz_fft / n (backward mean),
// 2: fo[b,c,:] = conj(wbar_c) fz / n (x = grad z, sampling)
template <typename T>
__global__ __launch_bounds__(kBlock) void grad_spectral_kernel(int mode, int d, size_t nh, double inv_n,
                                                               const typename Cx<T>::type* __restrict__ wbar,
                                                               const double* __restrict__ s2,
                                                               const double* __restrict__ az_d,
                                                               const double* __restrict__ ax_d,
                                                               const typename Cx<T>::type* __restrict__ fz,
                                                               const typename Cx<T>::type* __restrict__ fx,
                                                               typename Cx<T>::type* __restrict__ fo) {
    using C = typename Cx<T>::type;
    const size_t b = blockIdx.y;
    const double az = (mode == 2) ? 0.0 : az_d[b];
    const double ax = (mode == 2) ? 0.0 : ax_d[b];
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nh; i += (size_t)gridDim.x * kBlock) {
        C rz = fz[b * nh + i];
        if (mode != 2) {
            for (int c = 0; c < d; ++c) {
                const C w = wbar[(size_t)c * nh + i];
                const C x = fx[(b * d + c) * nh + i];
                rz.x += w.x * x.x - w.y * x.y;                         // :67 w_fft_bar * bx_fft, summed over axis 0
                rz.y += w.x * x.y + w.y * x.x;
            }
            const T res = (T)(inv_n / (az + ax * s2[i]));              // :66
            rz.x *= res;
            rz.y *= res;
        } else {
            rz.x *= (T)inv_n;
            rz.y *= (T)inv_n;
        }
        if (mode == 1) fo[b * nh + i] = rz;
        else
            for (int c = 0; c < d; ++c) {
                const C w = wbar[(size_t)c * nh + i];
                C o;
                o.x = w.x * rz.x + w.y * rz.y;                         // w_fft = conj(w_fft_bar), :78
                o.y = w.x * rz.y - w.y * rz.x;
                fo[(b * d + c) * nh + i] = o;
            }
    }
}

int grad_reserve(tramp_ctx* ctx, tramp_grad* g, size_t batch) {
    if (batch <= g->gwork_batch) return TRAMP_OK;
    if (ctx->capturing) return tramp_fail(ctx, TRAMP_EINVAL, "tramp_grad", "workspace growth inside graph capture");
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (g->gwork) (void)hipFree(g->gwork);
    g->gwork = nullptr;
    g->gwork_batch = 0;
    const tramp_conv* cv = &g->base;
    const size_t csz = 2 * dtype_size(cv->dtype);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t per = up(batch * cv->nh * csz);
    const size_t perd = up(batch * g->d * cv->nh * csz);
    const size_t rper = up(batch * g->d * cv->n * dtype_size(cv->dtype));
    const size_t scal = up(batch * sizeof(double));
    const size_t part = up(batch * (size_t)cv->nblk * sizeof(double));
    TRAMP_HIP_CHECK(ctx, hipMalloc(&g->gwork, per + 2 * perd + rper + scal + part));
    char* p = (char*)g->gwork;
    g->gfz = p;
    g->gfx = p + per;
    g->gfo = p + per + perd;
    g->gfr = p + per + 2 * perd;
    g->ga_raw = (double*)(p + per + 2 * perd + rper);
    g->gpartial = (double*)(p + per + 2 * perd + rper + scal);
    g->gwork_batch = batch;
    g->gproj_batch = 0;
    return TRAMP_OK;
}

// transforms of `count` signals of the channel's shape (count = batch or batch * d): own engine or rocFFT
int grad_forward(tramp_ctx* ctx, tramp_grad* g, size_t count, const void* in, void* spec) {
    tramp_conv* cv = &g->base;
    if (cv->fftn) return fftnd_forward(ctx, cv->fftn, count, in, spec);
    ConvPlans* p = nullptr;
    int rc = plans_for(ctx, cv, count, &p);
    if (rc) return rc;
    return fft_exec(ctx, p->fwd, p->info_fwd, in, spec);
}

int grad_inverse(tramp_ctx* ctx, tramp_grad* g, size_t count, void* spec, void* out) {
    tramp_conv* cv = &g->base;
    if (cv->fftn) return fftnd_inverse(ctx, cv->fftn, count, spec, out);
    ConvPlans* p = nullptr;
    int rc = plans_for(ctx, cv, count, &p);
    if (rc) return rc;
    return fft_exec(ctx, p->inv, p->info_inv, spec, out);
}

template <typename T>
void launch_grad_spectral(tramp_ctx* ctx, tramp_grad* g, int mode, size_t batch, const double* az_d,
                          const double* ax_d) {
    using C = typename Cx<T>::type;
    const tramp_conv* cv = &g->base;
    if (cv->fftn) {
        launch_spectral_nd<T>(ctx, cv, mode, batch, az_d, ax_d, g->gfz, g->gfx, g->gfo);
        return;
    }
    hipLaunchKernelGGL((grad_spectral_kernel<T>), dim3(grid_for(cv->nh), (unsigned)batch), dim3(kBlock), 0, ctx->stream,
                       mode, g->d, cv->nh, 1.0 / cv->n, (const C*)g->wbar_all, cv->s2, az_d, ax_d, (const C*)g->gfz,
                       (const C*)g->gfx, (C*)g->gfo);
}

}  // namespace

extern "C" int tramp_grad_destroy(tramp_ctx* ctx, tramp_grad* g) {
    if (!g) return TRAMP_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (auto& p : g->base.plans) {
        (void)rocfft_execution_info_destroy(p.info_fwd);
        (void)rocfft_execution_info_destroy(p.info_inv);
        (void)rocfft_plan_destroy(p.fwd);
        (void)rocfft_plan_destroy(p.inv);
        if (p.work_fwd) (void)hipFree(p.work_fwd);
        if (p.work_inv) (void)hipFree(p.work_inv);
    }
    if (g->gwork) (void)hipFree(g->gwork);
    if (g->wbar_all) (void)hipFree(g->wbar_all);
    if (g->base.s2) (void)hipFree(g->base.s2);
    conv_free_fft2(&g->base);
    delete g;
    return TRAMP_OK;
}

extern "C" int tramp_grad_create(tramp_ctx* ctx, int dtype, int ndim, const size_t* dims, const double* filters_d,
                                 tramp_grad** grad) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, grad && dims && filters_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F64 || dtype == TRAMP_F32);
    TRAMP_REQUIRE(ctx, ndim >= 1 && ndim <= 3);
    std::call_once(g_rocfft_once, [] { (void)rocfft_setup(); });
    tramp_grad* g = new tramp_grad();
    tramp_conv* cv = &g->base;
    g->d = ndim;
    g->wbar_all = g->gwork = nullptr;
    g->gwork_batch = 0;
    g->gproj_batch = 0;
    cv->dtype = dtype;
    cv->ndim = ndim;
    cv->n = 1;
    for (int a = 0; a < ndim; ++a) {
        if (dims[a] == 0) {
            delete g;
            return tramp_fail(ctx, TRAMP_EINVAL, "tramp_grad_create", "zero-length axis");
        }
        cv->dims[a] = dims[a];
        cv->n *= dims[a];
    }
    cv->nlast = dims[ndim - 1];
    cv->nhl = cv->nlast / 2 + 1;
    cv->nh = cv->n / cv->nlast * cv->nhl;
    cv->nblk = (int)((cv->nh + 4095) / 4096 > 256 ? 256 : (cv->nh + 4095) / 4096);
    cv->work = nullptr;
    cv->work_batch = cv->proj_batch = 0;
    cv->wbar = nullptr;
    cv->s2 = nullptr;
    cv->fft2 = nullptr;
    cv->fftn = nullptr;
    cv->cplx = false;
    cv->hl = 0;
    cv->ncomp = ndim;
    cv->wbar_nq = nullptr;
    cv->s2_nq = nullptr;
    cv->h = 0;
    cv->wbar_p = nullptr;
    cv->s2_p = nullptr;
    cv->partner = nullptr;
    cv->leaders = nullptr;
    cv->nlead = 0;
    cv->wk = nullptr;
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void* pwork = nullptr;
    double2* w64 = nullptr;
    auto cleanup = [&](bool failed) {
        if (w64) (void)hipFree(w64);
        if (pwork) (void)hipFree(pwork);
        if (info) (void)rocfft_execution_info_destroy(info);
        if (plan) (void)rocfft_plan_destroy(plan);
        if (failed) (void)tramp_grad_destroy(nullptr, g);
    };
    int rc = make_plan(ctx, cv, true, true, (size_t)ndim, &plan, &info, &pwork);
    if (rc) {
        cleanup(true);
        return rc;
    }
    const size_t csz = 2 * dtype_size(dtype);
    hipError_t e = hipMalloc(&w64, (size_t)ndim * cv->nh * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc(&g->wbar_all, (size_t)ndim * cv->nh * csz);
    if (e == hipSuccess) e = hipMalloc(&cv->s2, cv->nh * sizeof(double));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        cleanup(true);
        return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_grad_create", hipGetErrorString(e));
    }
    rc = fft_exec(ctx, plan, info, filters_d, w64);
    if (rc) {
        cleanup(true);
        return rc;
    }
    const unsigned gr = (unsigned)((cv->nh + kBlock - 1) / kBlock);
    if (dtype == TRAMP_F64)
        hipLaunchKernelGGL((grad_setup_kernel<double>), dim3(gr), dim3(kBlock), 0, ctx->stream, w64, ndim, cv->nh,
                           (double2*)g->wbar_all, cv->s2);
    else
        hipLaunchKernelGGL((grad_setup_kernel<float>), dim3(gr), dim3(kBlock), 0, ctx->stream, w64, ndim, cv->nh,
                           (float2*)g->wbar_all, cv->s2);
    std::vector<double> s2h(cv->nh);
    e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(s2h.data(), cv->s2, cv->nh * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        cleanup(true);
        return tramp_fail(ctx, TRAMP_EHIP, "tramp_grad_create", hipGetErrorString(e));
    }
    long double acc = 0.0L;
    for (size_t i = 0; i < cv->nh; ++i) {
        const size_t j = i % cv->nhl;
        acc += ((j == 0 || 2 * j == cv->nlast) ? 1.0L : 2.0L) * (long double)s2h[i];
    }
    cv->s_mean = (double)(acc / (long double)cv->n);
    rc = conv_setup_fftn(ctx, cv, ndim, w64, s2h);                     // own FFT engine when the shape allows
    if (rc) {
        cleanup(true);
        return rc;
    }
    cleanup(false);
    *grad = g;
    return TRAMP_OK;
}

extern "C" int tramp_grad_spectrum(tramp_ctx* ctx, const tramp_grad* g, double* s2_d, size_t* count, double* mean) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, g);
    return tramp_conv_spectrum(ctx, &g->base, s2_d, count, mean);
}

extern "C" int tramp_grad_variance(tramp_ctx* ctx, tramp_grad* g, int dir, size_t batch, const double* az_d,
                                   const double* ax_d, double* v_d, double* a_new_d, double* n_eff_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, g && az_d && ax_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    if (!batch) return TRAMP_OK;
    int rc = tramp_ws_reserve(ctx, batch * (size_t)g->base.nblk * sizeof(double));
    if (rc) return rc;
    return conv_scalars(ctx, &g->base, ctx->ws, dir, batch, az_d, ax_d, 0.0, nullptr, v_d, a_new_d, nullptr, n_eff_d,
                        (double)g->d);
}

extern "C" int tramp_grad_apply(tramp_ctx* ctx, tramp_grad* g, size_t batch, const void* z_d, void* x_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, g && z_d && x_d);
    if (!batch) return TRAMP_OK;
    int rc = grad_reserve(ctx, g, batch);
    if (rc) return rc;
    g->gproj_batch = 0;                                               // gfz is overwritten below
    rc = grad_forward(ctx, g, batch, z_d, g->gfz);
    if (rc) return rc;
    if (g->base.dtype == TRAMP_F64) launch_grad_spectral<double>(ctx, g, 2, batch, nullptr, nullptr);
    else launch_grad_spectral<float>(ctx, g, 2, batch, nullptr, nullptr);
    TRAMP_LAUNCH_CHECK(ctx);
    return grad_inverse(ctx, g, batch * g->d, g->gfo, x_d);
}

extern "C" int tramp_grad_message(tramp_ctx* ctx, tramp_grad* g, int dir, size_t batch, const double* az_d,
                                  const void* bz_d, const double* ax_d, const void* bx_d, double damping,
                                  const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d,
                                  void* r_d, double* v_d, int reuse) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, g && az_d && bz_d && ax_d && bx_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!batch) return TRAMP_OK;
    const void* work_before = g->gwork;
    int rc = grad_reserve(ctx, g, batch);
    if (rc) return rc;
    if (g->gwork != work_before || g->gproj_batch != batch) reuse = 0;
    g->gproj_batch = batch;
    tramp_conv* cv = &g->base;
    const size_t d = (size_t)g->d, n = cv->n, nh = cv->nh, esz = dtype_size(cv->dtype);
    {
        ProfScope ps(ctx, PROF_LIN_SCALARS, 0, 0);
        rc = conv_scalars(ctx, cv, g->gpartial, dir, batch, az_d, ax_d, damping, a_old_d, v_d, a_new_d, g->ga_raw,
                          nullptr, (double)g->d);
        if (rc) return rc;
    }
    {
        // transforms of the incoming messages, shared by the two directions of one iteration (TRAMP_REUSE_T / _U)
        const double cnt = ((reuse & TRAMP_REUSE_T) ? 0.0 : 1.0) + ((reuse & TRAMP_REUSE_U) ? 0.0 : (double)d);
        ProfScope ps(ctx, PROF_FFT, (double)batch * cnt * (n * esz + nh * 2 * esz), 0);
        if (!(reuse & TRAMP_REUSE_T)) {
            rc = grad_forward(ctx, g, batch, bz_d, g->gfz);
            if (rc) return rc;
        }
        if (!(reuse & TRAMP_REUSE_U)) {
            rc = grad_forward(ctx, g, batch * d, bx_d, g->gfx);
            if (rc) return rc;
        }
    }
    {
        ProfScope ps(ctx, PROF_LIN_SCALARS, (double)batch * nh * (2 * d + 2) * 2 * esz, 0);
        const int mode = (dir == TRAMP_FWD) ? 0 : 1;
        if (cv->dtype == TRAMP_F64) launch_grad_spectral<double>(ctx, g, mode, batch, az_d, ax_d);
        else launch_grad_spectral<float>(ctx, g, mode, batch, az_d, ax_d);
        TRAMP_LAUNCH_CHECK(ctx);
    }
    const bool fwd = dir == TRAMP_FWD;
    const size_t len = fwd ? d * n : n;                               // target vector length per instance
    const double* a_in = fwd ? ax_d : az_d;
    const void* b_in = fwd ? bx_d : bz_d;
    if (cv->fftn) {
        // own engine: the last pass of the inverse stores the message update (d signals per instance going forward:
        // the scalar tables are indexed by instance = signal / d, which rows_per_inst folds in)
        ProfScope ps(ctx, PROF_FFT, (double)batch * (fwd ? d : 1) * (n * esz * (damping != 0.0 ? 3 : 2) + nh * 2 * esz), 0);
        Fft2Epi epi{a_in, g->ga_raw, b_in, damping, b_old_d, b_new_d, r_d, PriorHintDev{}, 0};
        return fftnd_inverse_signals(ctx, cv->fftn, batch, fwd ? d : 1, g->gfo, &epi);
    }
    void* mean = r_d ? r_d : (damping != 0.0 ? g->gfr : b_new_d);
    {
        ProfScope ps(ctx, PROF_FFT, (double)batch * (fwd ? d : 1) * (n * esz + nh * 2 * esz), 0);
        rc = grad_inverse(ctx, g, fwd ? batch * d : batch, g->gfo, mean);
        if (rc) return rc;
    }
    {
        ProfScope ps(ctx, PROF_EW, (double)batch * len * esz * (damping != 0.0 ? 4 : 3), 0);
        dim3 gr(grid_for(len), (unsigned)batch);
        if (cv->dtype == TRAMP_F64)
            hipLaunchKernelGGL((conv_epilogue_kernel<double>), gr, dim3(kBlock), 0, ctx->stream, len, a_in, g->ga_raw,
                               (const double*)b_in, damping, (const double*)b_old_d, (const double*)mean,
                               (double*)b_new_d, (double*)nullptr);
        else
            hipLaunchKernelGGL((conv_epilogue_kernel<float>), gr, dim3(kBlock), 0, ctx->stream, len, a_in, g->ga_raw,
                               (const float*)b_in, damping, (const float*)b_old_d, (const float*)mean,
                               (float*)b_new_d, (float*)nullptr);
        TRAMP_LAUNCH_CHECK(ctx);
    }
    return TRAMP_OK;
}

// ---------------------------------------------------------------- DFT channel x = F z / sqrt(n)
// DFTChannel (tramp/channels/linear/dft_channel.py:7-77, real=True): a unitary change of basis,
// so the messages just travel through it: forward (ax_new, bx_new) = (az, F bz), backward
// (az_new, bz_new) = (ax, Re F^-1 bx), both with norm="ortho".  A complex vector is the planar
// real array [2, n] (row 0 real parts, row 1 imaginary parts) as in the reference.
struct DftPlans {
    size_t batch;
    rocfft_plan fwd, inv;
    rocfft_execution_info info_fwd, info_inv;
    void* work_fwd;
    void* work_inv;
};

struct tramp_dft {
    int dtype;
    int ndim;
    size_t dims[3];
    size_t n;
    std::vector<DftPlans> plans;
    void* work;          // two interleaved complex buffers [batch, n]
    size_t work_batch;
    void* c0;
    void* c1;
    // own FFT engine on the full complex array in its natural-order mode (fft_pow2.h: fftnd_forward_nat / _inverse_nat)
    FftNdPlan* fftn;
    bool complex_in;     // real=False: z is complex too (planar [2, n] on both sides)
};

namespace {

// real [batch, n] -> interleaved complex (imag 0)
template <typename T>
__global__ __launch_bounds__(kBlock) void dft_pack_real_kernel(size_t count, const T* __restrict__ in,
                                                               typename Cx<T>::type* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock) {
        typename Cx<T>::type c;
        c.x = in[i];
        c.y = T(0);
        out[i] = c;
    }
}

// planar [batch, 2, n] -> interleaved [batch, n]; with `idx`, position i takes element idx[i] (the transform's
// permuted order)
template <typename T>
__global__ __launch_bounds__(kBlock) void dft_pack_planar_kernel(size_t n, const T* __restrict__ in,
                                                                 typename Cx<T>::type* __restrict__ out,
                                                                 const int* __restrict__ idx) {
    const size_t b = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const size_t j = idx ? (size_t)idx[i] : i;
        typename Cx<T>::type c;
        c.x = in[b * 2 * n + j];
        c.y = in[b * 2 * n + n + j];
        out[b * n + i] = c;
    }
}

// interleaved -> message: planar [batch, 2, n] (FWD) or real part [batch, n] (BWD), scaled by
// 1/sqrt(n), constant damping fused
template <typename T>
__global__ __launch_bounds__(kBlock) void dft_unpack_kernel(size_t n, int planar, T scale,
                                                            const typename Cx<T>::type* __restrict__ in,
                                                            double damping, const T* b_old, T* out,
                                                            const int* __restrict__ idx) {
    const size_t b = blockIdx.y;
    const T d = T(damping), omd = T(1.0 - damping);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const typename Cx<T>::type c = in[b * n + (idx ? (size_t)idx[i] : i)];   // idx: where frequency i sits
        if (planar) {
            T re = c.x * scale, im = c.y * scale;
            const size_t o = b * 2 * n + i;
            if (damping != 0.0) {
                re = d * b_old[o] + omd * re;
                im = d * b_old[o + n] + omd * im;
            }
            out[o] = re;
            out[o + n] = im;
        } else {
            T re = c.x * scale;
            if (damping != 0.0) re = d * b_old[b * n + i] + omd * re;
            out[b * n + i] = re;
        }
    }
}

__global__ void dft_scalar_kernel(size_t batch, const double* __restrict__ a_in, double damping,
                                  const double* __restrict__ a_old, double* __restrict__ a_new) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double a = a_in[b];
    a_new[b] = (damping != 0.0) ? damping * a_old[b] + (1.0 - damping) * a : a;
}

}  // namespace

extern "C" int tramp_dft_destroy(tramp_ctx* ctx, tramp_dft* h);

extern "C" int tramp_dft_create(tramp_ctx* ctx, int dtype, int ndim, const size_t* dims, tramp_dft** dft) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, dft && dims);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F64 || dtype == TRAMP_F32);
    TRAMP_REQUIRE(ctx, ndim >= 1 && ndim <= 3);
    std::call_once(g_rocfft_once, [] { (void)rocfft_setup(); });
    tramp_dft* h = new tramp_dft();
    h->dtype = dtype;
    h->ndim = ndim;
    h->n = 1;
    for (int d = 0; d < ndim; ++d) {
        if (dims[d] == 0) {
            delete h;
            return tramp_fail(ctx, TRAMP_EINVAL, "tramp_dft_create", "zero-length axis");
        }
        h->dims[d] = dims[d];
        h->n *= dims[d];
    }
    h->work = nullptr;
    h->work_batch = 0;
    h->fftn = nullptr;
    h->complex_in = false;
    // The spectrum is this channel's MESSAGE, so it has to leave in natural order.  The own engine delivers that at no extra
    // pass (fft_pow2.h natural-order mode: last axis <= 4096 complex): level with rocFFT at 1024 x 1024 (1.04x f64, 0.96x
    // f32), behind it at 4096 x 4096 (0.86x: the second pass of a long axis must go out of place) and on small arrays
    // (0.6-0.8x, launch-bound) - tools/bench_dft.py, profiles/r03/bench_dft_*.txt.  Round 5: the own engine is the default
    // where it applies (no vendor library on the iteration path); TRAMP_DFT_FFT=rocfft selects the library transform.
    const char* eng = getenv("TRAMP_DFT_FFT");
    if (!(eng && strcmp(eng, "rocfft") == 0) && fftnd_supported(dtype, ndim, h->dims, false)) {
        int rc = fftnd_plan_create(ctx, dtype, ndim, h->dims, &h->fftn, false);
        if (!rc) {
            if (fftnd_natural_ok(h->fftn)) rc = fftnd_enable_natural(ctx, h->fftn);
            else {
                fftnd_plan_destroy(h->fftn);
                h->fftn = nullptr;
            }
        }
        if (rc) {
            (void)tramp_dft_destroy(ctx, h);
            return rc;
        }
    }
    *dft = h;
    return TRAMP_OK;
}

// real=False (dft_channel.py:31-45): z is complex as well, carried as the planar array [2, n] like x
extern "C" int tramp_dft_set_complex_input(tramp_ctx* ctx, tramp_dft* h, int on) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, h);
    h->complex_in = on != 0;
    return TRAMP_OK;
}

extern "C" int tramp_dft_destroy(tramp_ctx* ctx, tramp_dft* h) {
    if (!h) return TRAMP_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (auto& p : h->plans) {
        (void)rocfft_execution_info_destroy(p.info_fwd);
        (void)rocfft_execution_info_destroy(p.info_inv);
        (void)rocfft_plan_destroy(p.fwd);
        (void)rocfft_plan_destroy(p.inv);
        if (p.work_fwd) (void)hipFree(p.work_fwd);
        if (p.work_inv) (void)hipFree(p.work_inv);
    }
    if (h->work) (void)hipFree(h->work);
    if (h->fftn) fftnd_plan_destroy(h->fftn);
    delete h;
    return TRAMP_OK;
}

extern "C" int tramp_dft_message(tramp_ctx* ctx, tramp_dft* h, int dir, size_t batch, const double* a_in_d,
                                 const void* b_in_d, double damping, const double* a_old_d, const void* b_old_d,
                                 double* a_new_d, void* b_new_d) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, h && a_in_d && b_in_d && a_new_d && b_new_d);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    TRAMP_REQUIRE(ctx, damping == 0.0 || (a_old_d && b_old_d));
    if (!batch) return TRAMP_OK;
    const size_t n = h->n, esz = dtype_size(h->dtype);
    if (batch > h->work_batch) {
        if (ctx->capturing) return tramp_fail(ctx, TRAMP_EINVAL, "tramp_dft", "workspace growth inside graph capture");
        TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (h->work) (void)hipFree(h->work);
        h->work = nullptr;
        h->work_batch = 0;
        const size_t per = ((batch * n * 2 * esz + 255) / 256) * 256;
        TRAMP_HIP_CHECK(ctx, hipMalloc(&h->work, 2 * per));
        h->c0 = h->work;
        h->c1 = (char*)h->work + per;
        h->work_batch = batch;
    }
    DftPlans* pl = nullptr;
    for (auto& p : h->plans)
        if (p.batch == batch) pl = &p;
    if (!pl && !h->fftn) {
        if (ctx->capturing) return tramp_fail(ctx, TRAMP_EINVAL, "tramp_dft", "FFT plan creation inside graph capture");
        DftPlans p{};
        p.batch = batch;
        const bool f64 = h->dtype == TRAMP_F64;
        int rc = make_plan_dims(ctx, h->ndim, h->dims, rocfft_transform_type_complex_forward, f64, batch, &p.fwd,
                                &p.info_fwd, &p.work_fwd);
        if (rc) return rc;
        rc = make_plan_dims(ctx, h->ndim, h->dims, rocfft_transform_type_complex_inverse, f64, batch, &p.inv,
                            &p.info_inv, &p.work_inv);
        if (rc) return rc;
        h->plans.push_back(p);
        pl = &h->plans.back();
    }
    const bool fwd = dir == TRAMP_FWD;
    const double scale = 1.0 / sqrt((double)n);
    dim3 g(grid_for(n), (unsigned)batch);
    hipLaunchKernelGGL(dft_scalar_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, ctx->stream, batch, a_in_d,
                       damping, a_old_d, a_new_d);
    const bool own = h->fftn != nullptr;
    const bool real_in = fwd && !h->complex_in;                 // the only real-valued input
    const bool planar_out = fwd || h->complex_in;
    const int* pack_idx = nullptr;                               // (index tables: unused since the natural-order passes)
    const int* unpack_idx = nullptr;
    void* spec = h->c1;
    {
        ProfScope ps(ctx, PROF_EW, (double)batch * n * esz * (real_in ? 3 : 4), 0);
        if (h->dtype == TRAMP_F64) {
            if (real_in)
                hipLaunchKernelGGL((dft_pack_real_kernel<double>), dim3(grid_for(batch * n)), dim3(kBlock), 0,
                                   ctx->stream, batch * n, (const double*)b_in_d, (double2*)h->c0);
            else
                hipLaunchKernelGGL((dft_pack_planar_kernel<double>), g, dim3(kBlock), 0, ctx->stream, n,
                                   (const double*)b_in_d, (double2*)h->c0, pack_idx);
        } else {
            if (real_in)
                hipLaunchKernelGGL((dft_pack_real_kernel<float>), dim3(grid_for(batch * n)), dim3(kBlock), 0,
                                   ctx->stream, batch * n, (const float*)b_in_d, (float2*)h->c0);
            else
                hipLaunchKernelGGL((dft_pack_planar_kernel<float>), g, dim3(kBlock), 0, ctx->stream, n,
                                   (const float*)b_in_d, (float2*)h->c0, pack_idx);
        }
        TRAMP_LAUNCH_CHECK(ctx);
    }
    {
        ProfScope ps(ctx, PROF_FFT, (double)batch * n * 4 * esz, 0);
        int rc;
        if (own) rc = fwd ? fftnd_forward_nat(ctx, h->fftn, batch, h->c0, h->c1, &spec)
                          : fftnd_inverse_nat(ctx, h->fftn, batch, h->c0, h->c1, &spec);
        else rc = fwd ? fft_exec(ctx, pl->fwd, pl->info_fwd, h->c0, h->c1) : fft_exec(ctx, pl->inv, pl->info_inv, h->c0, h->c1);
        if (rc) return rc;
    }
    {
        ProfScope ps(ctx, PROF_EW, (double)batch * n * esz * (planar_out ? 4 : 3), 0);
        if (h->dtype == TRAMP_F64)
            hipLaunchKernelGGL((dft_unpack_kernel<double>), g, dim3(kBlock), 0, ctx->stream, n, planar_out ? 1 : 0, scale,
                               (const double2*)spec, damping, (const double*)b_old_d, (double*)b_new_d, unpack_idx);
        else
            hipLaunchKernelGGL((dft_unpack_kernel<float>), g, dim3(kBlock), 0, ctx->stream, n, planar_out ? 1 : 0,
                               (float)scale, (const float2*)spec, damping, (const float*)b_old_d, (float*)b_new_d,
                               unpack_idx);
        TRAMP_LAUNCH_CHECK(ctx);
    }
    return TRAMP_OK;
}

// ---------------------------------------------------------------- complex convolution channel (real=False)
// ConvChannel(filter, real=False) (tramp/channels/linear/conv_channel.py:14-32, 81-109): x, w, z complex, every vector the
// planar real array [2, n] (real parts, then imaginary parts).  The spectra are full complex transforms of n points and
// the update is purely elementwise in the Fourier basis - no Hermitian pairing.  Own FFT engine on the complex array when
// the shape allows (tables stored in the transform's permuted order, so the elementwise kernel is the natural-order one),
// rocFFT complex plans otherwise.  Scalars: n_eff is the plain mean over the n bins (conv_channel.py:69-79).
namespace {

// interleaved [batch, n] -> planar [batch, 2, n]; with a_in: message update b_new = r (a_in + a_raw) - b_in on the way
template <typename T>
__global__ __launch_bounds__(kBlock) void cplx_unpack_kernel(size_t n, const typename Cx<T>::type* __restrict__ c,
                                                             const double* __restrict__ a_in,
                                                             const double* __restrict__ a_raw, const T* __restrict__ b_in,
                                                             double damping, const T* b_old, T* out, T* r_out) {
    const size_t b = blockIdx.y;
    const T asum = a_in ? (T)(a_in[b] + a_raw[b]) : T(1);
    const T d = (T)damping, omd = (T)(1.0 - damping);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const typename Cx<T>::type v = c[b * n + i];
        const size_t o = b * 2 * n + i;
        T re = v.x, im = v.y;
        if (r_out) {
            r_out[o] = re;
            r_out[o + n] = im;
        }
        if (a_in) {
            re = re * asum - b_in[o];
            im = im * asum - b_in[o + n];
            if (damping != 0.0) {
                re = d * b_old[o] + omd * re;
                im = d * b_old[o + n] + omd * im;
            }
        }
        out[o] = re;
        out[o + n] = im;
    }
}

template <typename T>
int conv_complex_run_t(tramp_ctx* ctx, tramp_conv* cv, int mode, size_t batch, const double* az_d, const void* bz_d,
                       const double* ax_d, const void* bx_d, double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d, int reuse) {
    using C = typename Cx<T>::type;
    const void* work_before = cv->work;
    int rc = conv_reserve(ctx, cv, batch);
    if (rc) return rc;
    if (cv->work != work_before || cv->proj_batch != batch || mode == 2) reuse = 0;
    cv->proj_batch = (mode == 2) ? 0 : batch;
    ConvPlans* p = nullptr;
    if (!cv->fftn) {
        rc = plans_for(ctx, cv, batch, &p);
        if (rc) return rc;
    }
    const size_t n = cv->n, esz = dtype_size(cv->dtype);
    const dim3 g(grid_for(n), (unsigned)batch);
    if (mode != 2) {
        ProfScope ps(ctx, PROF_LIN_SCALARS, 0, 0);
        rc = conv_scalars(ctx, cv, cv->partial, mode == 0 ? TRAMP_FWD : TRAMP_BWD, batch, az_d, ax_d, damping, a_old_d, v_d,
                          a_new_d, cv->a_raw, nullptr);
        if (rc) return rc;
    }
    // planar -> interleaved (staged in fo) -> spectrum
    auto forward = [&](const void* src, void* spec) -> int {
        hipLaunchKernelGGL((dft_pack_planar_kernel<T>), g, dim3(kBlock), 0, ctx->stream, n, (const T*)src, (C*)cv->fo,
                           (const int*)nullptr);
        ProfScope ps(ctx, PROF_FFT, (double)batch * n * 4 * esz, 0);
        return conv_forward(ctx, cv, p, batch, cv->fo, spec);
    };
    if (!(reuse & TRAMP_REUSE_T)) {
        rc = forward(bz_d, cv->fz);
        if (rc) return rc;
    }
    if (mode != 2 && !(reuse & TRAMP_REUSE_U)) {
        rc = forward(bx_d, cv->fx);
        if (rc) return rc;
    }
    {
        ProfScope ps(ctx, PROF_LIN_SCALARS, (double)batch * n * 6 * esz, 0);
        hipLaunchKernelGGL((conv_spectral_kernel<T>), g, dim3(kBlock), 0, ctx->stream, mode, n, 1.0 / n, (const C*)cv->wbar,
                           cv->s2_p ? cv->s2_p : cv->s2, az_d, ax_d, (const C*)cv->fz, (const C*)cv->fx, (C*)cv->fo);
        TRAMP_LAUNCH_CHECK(ctx);
    }
    {
        ProfScope ps(ctx, PROF_FFT, (double)batch * n * 4 * esz, 0);
        rc = conv_inverse(ctx, cv, p, batch, cv->fo, cv->fzc);            // interleaved mean
        if (rc) return rc;
    }
    {
        ProfScope ps(ctx, PROF_EW, (double)batch * n * 2 * esz * (damping != 0.0 ? 4 : 3), 0);
        if (mode == 2)
            hipLaunchKernelGGL((cplx_unpack_kernel<T>), g, dim3(kBlock), 0, ctx->stream, n, (const C*)cv->fzc,
                               (const double*)nullptr, (const double*)nullptr, (const T*)nullptr, 0.0, (const T*)nullptr,
                               (T*)b_new_d, (T*)nullptr);
        else
            hipLaunchKernelGGL((cplx_unpack_kernel<T>), g, dim3(kBlock), 0, ctx->stream, n, (const C*)cv->fzc,
                               mode == 0 ? ax_d : az_d, (const double*)cv->a_raw, (const T*)(mode == 0 ? bx_d : bz_d), damping,
                               (const T*)b_old_d, (T*)b_new_d, (T*)r_d);
        TRAMP_LAUNCH_CHECK(ctx);
    }
    return TRAMP_OK;
}

}  // namespace

int conv_complex_run(tramp_ctx* ctx, tramp_conv* cv, int mode, size_t batch, const double* az_d, const void* bz_d,
                     const double* ax_d, const void* bx_d, double damping, const double* a_old_d, const void* b_old_d,
                     double* a_new_d, void* b_new_d, void* r_d, double* v_d, int reuse) {
    return cv->dtype == TRAMP_F64
               ? conv_complex_run_t<double>(ctx, cv, mode, batch, az_d, bz_d, ax_d, bx_d, damping, a_old_d, b_old_d, a_new_d,
                                            b_new_d, r_d, v_d, reuse)
               : conv_complex_run_t<float>(ctx, cv, mode, batch, az_d, bz_d, ax_d, bx_d, damping, a_old_d, b_old_d, a_new_d,
                                           b_new_d, r_d, v_d, reuse);
}

// filter_d: planar [2, n] f64 (real parts, imaginary parts)
extern "C" int tramp_conv_create_complex(tramp_ctx* ctx, int dtype, int ndim, const size_t* dims, const double* filter_d,
                                         tramp_conv** conv) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, conv && dims && filter_d);
    TRAMP_REQUIRE(ctx, dtype == TRAMP_F64 || dtype == TRAMP_F32);
    TRAMP_REQUIRE(ctx, ndim >= 1 && ndim <= 3);
    std::call_once(g_rocfft_once, [] { (void)rocfft_setup(); });
    tramp_conv* cv = new tramp_conv();
    cv->dtype = dtype;
    cv->ndim = ndim;
    cv->n = 1;
    for (int d = 0; d < ndim; ++d) {
        if (dims[d] == 0) {
            delete cv;
            return tramp_fail(ctx, TRAMP_EINVAL, "tramp_conv_create_complex", "zero-length axis");
        }
        cv->dims[d] = dims[d];
        cv->n *= dims[d];
    }
    cv->cplx = true;
    cv->nlast = dims[ndim - 1];
    cv->nhl = 1;                                  // every bin of the full spectrum counts once (herm_weight)
    cv->nh = cv->n;
    cv->nblk = (int)((cv->nh + 4095) / 4096 > 256 ? 256 : (cv->nh + 4095) / 4096);
    cv->ncomp = 1;
    cv->work = cv->wbar = cv->wbar_p = cv->wk = cv->wbar_nq = nullptr;
    cv->s2 = cv->s2_p = cv->s2_nq = nullptr;
    cv->partner = cv->leaders = nullptr;
    cv->fft2 = nullptr;
    cv->fftn = nullptr;
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void* pwork = nullptr;
    double2 *wi = nullptr, *w64 = nullptr;
    auto cleanup = [&](bool failed) {
        if (wi) (void)hipFree(wi);
        if (w64) (void)hipFree(w64);
        if (pwork) (void)hipFree(pwork);
        if (info) (void)rocfft_execution_info_destroy(info);
        if (plan) (void)rocfft_plan_destroy(plan);
        if (failed) (void)tramp_conv_destroy(ctx, cv);
    };
    const size_t n = cv->n, csz = 2 * dtype_size(dtype);
    int rc = make_plan(ctx, cv, true, true, 1, &plan, &info, &pwork);       // f64 complex forward, setup only
    if (rc) {
        cleanup(true);
        return rc;
    }
    hipError_t e = hipMalloc((void**)&wi, n * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void**)&w64, n * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc(&cv->wbar, n * csz);
    if (e == hipSuccess) e = hipMalloc((void**)&cv->s2, n * sizeof(double));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        cleanup(true);
        return tramp_fail(ctx, TRAMP_ENOMEM, "tramp_conv_create_complex", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((dft_pack_planar_kernel<double>), dim3(grid_for(n), 1), dim3(kBlock), 0, ctx->stream, n, filter_d, wi,
                       (const int*)nullptr);
    rc = fft_exec(ctx, plan, info, wi, w64);
    if (rc) {
        cleanup(true);
        return rc;
    }
    std::vector<double2> wh(n);
    e = hipMemcpyAsync(wh.data(), w64, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        cleanup(true);
        return tramp_fail(ctx, TRAMP_EHIP, "tramp_conv_create_complex", hipGetErrorString(e));
    }
    // own engine: the tables go into the transform's permuted order (the spectral kernel is elementwise)
    std::vector<double2> wt(wh);
    if (own_engine_wanted(n) && fftnd_supported(dtype, ndim, cv->dims, false)) {
        rc = fftnd_plan_create(ctx, dtype, ndim, cv->dims, &cv->fftn, false);
        if (rc) {
            cleanup(true);
            return rc;
        }
        cv->h = n;
        std::vector<size_t> kf[3];
        for (int a = 0; a < ndim; ++a) {
            kf[a].resize(cv->dims[a]);
            for (size_t i = 0; i < cv->dims[a]; ++i) kf[a][i] = fftnd_kfreq(cv->fftn, a, i);
        }
        for (size_t p = 0; p < n; ++p) {
            size_t rem = p, k = 0, stride = 1;
            for (int a = ndim - 1; a >= 0; --a) {
                k += kf[a][rem % cv->dims[a]] * stride;
                rem /= cv->dims[a];
                stride *= cv->dims[a];
            }
            wt[p] = wh[k];
        }
    }
    std::vector<double> s2h(n);
    long double acc = 0.0L;
    for (size_t i = 0; i < n; ++i) {
        s2h[i] = wt[i].x * wt[i].x + wt[i].y * wt[i].y;                    // conv_channel.py:45
        acc += (long double)s2h[i];
    }
    cv->s_mean = (double)(acc / (long double)n);
    if (dtype == TRAMP_F64) e = hipMemcpy(cv->wbar, wt.data(), n * csz, hipMemcpyHostToDevice);
    else {
        std::vector<float2> wf(n);
        for (size_t i = 0; i < n; ++i) wf[i] = float2{(float)wt[i].x, (float)wt[i].y};
        e = hipMemcpy(cv->wbar, wf.data(), n * csz, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(cv->s2, s2h.data(), n * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        cleanup(true);
        return tramp_fail(ctx, TRAMP_EHIP, "tramp_conv_create_complex", hipGetErrorString(e));
    }
    cleanup(false);
    *conv = cv;
    return TRAMP_OK;
}
