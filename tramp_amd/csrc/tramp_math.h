// tramp_math.h - per-component belief arithmetic of the Tree-AMP hot path.
//
// Scalar building blocks shared by every elementwise HIP kernel; templated on
// the vector element type T (float / double).  Each function names the
// reference lines it restates (paths relative to sphinxteam/tramp).  The same
// header is compiled for the host by tests/host_math (a unit-test harness for
// the arithmetic, never a product path) - hence the TM_HD macro.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TM_HD __host__ __device__ __forceinline__
#define TM_TABLE __device__ const            // coefficient tables: device globals (staged into LDS by the kernels)
#define TM_SCALAR_TABLE __constant__         // short coefficient lists read through the scalar cache
#else
#define TM_HD inline
#define TM_TABLE static const
#define TM_SCALAR_TABLE static const
#endif
// The arithmetic below is ONE set of expressions for the device and for the host harness (tests/host_math); what differs
// is confined to the primitives of namespace tmath::hw further down - the hardware's reciprocal / exp2 estimates, where
// a coefficient table is read from (LDS on the device, the table itself on the host), the wave-wide vote and the
// library's erfcx (glibc has none).
// true when the predicate holds on every active lane of the wave (device); the plain predicate elsewhere
#if defined(__HIP_DEVICE_COMPILE__)
#define TM_WAVE_ALL(p) (__all((p) ? 1 : 0) != 0)
#else
#define TM_WAVE_ALL(p) (p)
#endif

namespace tmath {

constexpr double kAMin = 1e-11;   // tramp/base.py:238-239
constexpr double kAMax = 1e+11;
constexpr double kVFloor = 1e-20; // tramp/base.py:44-46
constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kSqrt2 = 1.4142135623730951;          // np.sqrt(2)
constexpr double kSqrt2OverPi = 0.7978845608028654;    // np.sqrt(2/np.pi)
constexpr double kTwoOverSqrtPi = 1.1283791670955126;  // 2/np.sqrt(np.pi)

template <typename T> struct RV { T r, v; };

// ---- special functions -------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
TM_HD double erfcx_(double x) { return ::erfcx(x); }
TM_HD float erfcx_(float x) { return ::erfcxf(x); }
#else
// Host stand-in used only by the unit-test harness (glibc has no erfcx).
inline double erfcx_(double x) {
    if (x < 25.0) return ::exp(x * x) * ::erfc(x);
    double y = 1.0 / (2.0 * x * x);   // asymptotic series, rel. error < 1e-16 for x >= 25
    return (1.0 / (x * 1.772453850905516)) * (1.0 - y * (1.0 - 3.0 * y * (1.0 - 5.0 * y * (1.0 - 7.0 * y))));
}
inline float erfcx_(float x) { return (float)erfcx_((double)x); }
#endif

TM_HD double exp_(double x) { return ::exp(x); }
TM_HD float exp_(float x) { return ::expf(x); }
TM_HD double log_(double x) { return ::log(x); }
TM_HD float log_(float x) { return ::logf(x); }
TM_HD double sqrt_(double x) { return ::sqrt(x); }
TM_HD float sqrt_(float x) { return ::sqrtf(x); }
TM_HD double tanh_(double x) { return ::tanh(x); }
TM_HD float tanh_(float x) { return ::tanhf(x); }
TM_HD double fabs_(double x) { return ::fabs(x); }
TM_HD float fabs_(float x) { return ::fabsf(x); }
TM_HD double fmax_(double a, double b) { return ::fmax(a, b); }
TM_HD float fmax_(float a, float b) { return ::fmaxf(a, b); }
TM_HD double fmin_(double a, double b) { return ::fmin(a, b); }
TM_HD float fmin_(float a, float b) { return ::fminf(a, b); }
TM_HD double erf_(double x) { return ::erf(x); }
TM_HD float erf_(float x) { return ::erff(x); }
TM_HD double fma_(double a, double b, double c) { return ::fma(a, b, c); }
TM_HD float fma_(float a, float b, float c) { return ::fmaf(a, b, c); }

// erfcx for the streaming kernels (K4, K7).  The library's f64 erfcx costs 130-160 instructions on
// gfx950 (three range branches, all taken by some lane of a wave); here: erfcx(x) = t g(t), t = 4/(4 + |x|),
// g a degree-7 polynomial on one of 32 intervals of t (erfcx_table.h, fitted to mpmath, relative error
// 3e-15; f32: degree 4, own table), eight LDS reads (see below) and seven FMAs; x < 0 by
// erfcx(x) = 2 exp(x^2) - erfcx(-x).  No branch mask depends on an erfcx value.
}  // namespace tmath
#if defined(__HIPCC__)
#include "erfcx_table.h"
#include "erfcx_ttable.h"
#include "tanh_table.h"
#else
#define __device__              // the generated tables say `__device__ const`: plain constants for a host compiler
#include "erfcx_table.h"
#include "erfcx_ttable.h"
#include "tanh_table.h"
#undef __device__
#endif
namespace tmath {
#ifndef TRAMP_ERFCX_UNIFORM
#define TRAMP_ERFCX_UNIFORM 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// The table is read from LDS, coefficient-major ([j][k], 8-byte entries): rows 16..31 (x <= 4, i.e. all
// but the far tail) sit in 16 different bank pairs, so the 8 ds_read_b64 of one evaluation are
// conflict-free whatever rows the lanes of a wave need.  (Read straight from global memory the same
// lookups were 4 divergent 16-byte gathers per evaluation and the texture path, not the VALU, set the
// pace: 487 us for the 109-instruction sign likelihood against 280 us for the 163-instruction abs one.)
// Every kernel that evaluates erfcx_hot calls erfcx_lds_init() first (all threads; it ends in a barrier).
__shared__ double g_erfcx_lds[(kErfcxD + 1) * kErfcxK];
// Round 5: the second table holds ONE value per grid point, erfcx(x_k) at x_k = -7.5 + k / 128 (erfcx_ttable.h, 15.5 KiB
// of LDS; f32: k / 64, 3.9 KiB): erfcx' = 2 x erfcx - 2/sqrt(pi) gives every Taylor coefficient about x_k from that value by
// a three-term recurrence, so an evaluation is one 8-byte LDS read and ~30 VALU instructions, no reciprocal, and no
// exp(x^2) for negative arguments.  (Round 3's table held a degree-7 polynomial per interval of 1/32: eight reads per
// evaluation, rows colliding in the banks - the LDS, not the VALU, set the pace of the two-region channels K7:
// tools/probes/probe_k7.hip, 427 us against 358 us for the moments pass at 64 x 1e6.)  Taken when every lane of the wave
// is inside [-7.5, 8), else the whole wave evaluates the reciprocal-mapped table below.  Used by K7 (erfcx_hot<true>,
// pwl_moment_form).  -DTRAMP_ERFCX_UNIFORM=0 removes it.
#if TRAMP_ERFCX_UNIFORM
__shared__ double g_erfcx_tlds[kErfcxTK];
__shared__ float g_erfcx_tlds_f[kErfcxTKF];
#endif
// tanh(k / 64) (tanh_table.h, 9.8 KiB of LDS; f32: k / 32, 1.2 KiB): the binary prior's tanh_tab.  Staged by tanh_setup().
__shared__ double g_tanh_lds[kTanhK];
__shared__ float g_tanh_lds_f[kTanhKF];
template <bool UNI = false> __device__ __forceinline__ void erfcx_lds_init() {
    for (int i = threadIdx.x; i < (kErfcxD + 1) * kErfcxK; i += blockDim.x)
        g_erfcx_lds[i] = kErfcxTable[i % kErfcxK][i / kErfcxK];
#if TRAMP_ERFCX_UNIFORM
    if (UNI)
        for (int i = threadIdx.x; i < kErfcxTK; i += blockDim.x) g_erfcx_tlds[i] = kErfcxTTable[i];
#endif
    __syncthreads();
}
// f32 kernels: their own table (degree 4, 4-byte entries: the 32 rows sit in 32 different banks)
__shared__ float g_erfcx_lds_f[(kErfcxDF + 1) * kErfcxKF];
template <bool UNI = false> __device__ __forceinline__ void erfcx_lds_init_f() {
    for (int i = threadIdx.x; i < (kErfcxDF + 1) * kErfcxKF; i += blockDim.x)
        g_erfcx_lds_f[i] = kErfcxTableF[i % kErfcxKF][i / kErfcxKF];
#if TRAMP_ERFCX_UNIFORM
    if (UNI)
        for (int i = threadIdx.x; i < kErfcxTKF; i += blockDim.x) g_erfcx_tlds_f[i] = kErfcxTTableF[i];
#endif
    __syncthreads();
}
#endif

// ---- the primitives that differ between the device and the host harness ---------------------------------------------
namespace hw {
#if defined(__HIP_DEVICE_COMPILE__)
// v_rcp_f64 is good to 4.6e-8, v_rcp_f32 / v_exp_f32 to ~1 ulp of f32 (tools/probes/probe_rcp_ldexp.hip)
__device__ __forceinline__ double rcp_est(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ float rcp_est(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exp2_est(float x) { return __builtin_amdgcn_exp2f(x); }
// coefficient j of interval k: the LDS images staged by erfcx_lds_init (coefficient-major)
__device__ __forceinline__ double erfcx_coef(int j, int k) { return g_erfcx_lds[j * kErfcxK + k]; }
__device__ __forceinline__ float erfcx_coef_f(int j, int k) { return g_erfcx_lds_f[j * kErfcxKF + k]; }
#if TRAMP_ERFCX_UNIFORM
__device__ __forceinline__ double erfcx_tval(int k) { return g_erfcx_tlds[k]; }
__device__ __forceinline__ float erfcx_tval_f(int k) { return g_erfcx_tlds_f[k]; }
#endif
__device__ __forceinline__ double tanh_tval(int k) { return g_tanh_lds[k]; }
__device__ __forceinline__ float tanh_tval_f(int k) { return g_tanh_lds_f[k]; }
#else
// host harness: an f32 division stands for the f64 estimate (1e-7 instead of 4.6e-8: the cubic correction of rcp_1to2
// ends below 1e-20 either way), the correctly rounded f32 operations for the f32 estimates, the tables where they are
TM_HD double rcp_est(double x) {
    int e;
    const double m = ::frexp(x, &e);
    return ::ldexp((double)(1.0f / (float)m), -e);
}
TM_HD float rcp_est(float x) { return 1.0f / x; }
TM_HD float exp2_est(float x) { return ::exp2f(x); }
TM_HD double erfcx_coef(int j, int k) { return kErfcxTable[k][j]; }
TM_HD float erfcx_coef_f(int j, int k) { return kErfcxTableF[k][j]; }
#if TRAMP_ERFCX_UNIFORM
TM_HD double erfcx_tval(int k) { return kErfcxTTable[k]; }
TM_HD float erfcx_tval_f(int k) { return kErfcxTTableF[k]; }
#endif
TM_HD double tanh_tval(int k) { return kTanhTable[k]; }
TM_HD float tanh_tval_f(int k) { return kTanhTableF[k]; }
#endif
}  // namespace hw

// exp(x) for x <= 0 in f64 on the device: the library routine carries overflow handling and a two-level
// reduction it does not need here (~46 VALU instructions); n = rint(x log2 e), r = x - n ln2 (two parts),
// degree-12 Taylor polynomial on |r| <= ln2/2 (truncation 2e-16), one ldexp: ~22 instructions, 1-2 ulp.
// core: exp(x) for |x| within the f64 range (callers clamp); v_ldexp_f64 scales by any n in one step,
// denormal results included (tools/probes/probe_rcp_ldexp.hip)
// The Taylor coefficients are read through the scalar cache (uniform address: s_load into SGPRs, which a
// v_fma_f64 takes as an operand): as literals the compiler materialises each in a VGPR pair first - 13
// v_mov_b64 per evaluation on the VALU that these kernels are bound by.
TM_SCALAR_TABLE double kExpTaylor[13] = {
    2.08767569878680989792e-09,             // 1/12!
    2.50521083854417187751e-08, 2.75573192239858906526e-07, 2.75573192239858906526e-06, 2.48015873015873015873e-05,
    1.98412698412698412698e-04, 1.38888888888888888889e-03, 8.33333333333333333333e-03, 4.16666666666666666667e-02,
    1.66666666666666666667e-01, 0.5, 1.0, 1.0};
TM_HD double exp_core(double x) {
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = ::fma(n, -6.93147180369123816490e-01, x);
    r = ::fma(n, -1.90821492927058770002e-10, r);
    double p = kExpTaylor[0];
#pragma unroll
    for (int i = 1; i < 13; ++i) p = ::fma(p, r, kExpTaylor[i]);
    return ::ldexp(p, (int)n);
}
// any argument: the result saturates to 0 below -745.2 and to +inf above 709.78 like the library's
TM_HD double exp_fast(double x) {
    const double e = exp_core(x);
    return x < -745.2 ? 0.0 : (x > 709.79 ? __builtin_inf() : e);      // -inf -> 0, +inf -> inf; NaN propagates through p
}
TM_HD double exp_nonpos(double x) { return exp_fast(x); }
// Saturating by clamping the ARGUMENT (two instructions instead of two compare + select pairs); a NaN
// argument is swallowed by the clamp (maxNum / minNum), so these are for call sites whose result is
// multiplied into, or added to, a quantity that carries the NaN anyway.
TM_HD double exp_clamped(double x) { return exp_core(::fmin(::fmax(x, -746.0), 709.78)); }
TM_HD double exp_clamped_nonpos(double x) { return exp_core(::fmax(x, -746.0)); }
// f32: two-part reduction, v_exp_f32 on |r| <= ln2 / 2, one ldexp (~1e-7 relative; expf is ~15 instructions)
TM_HD float exp_core(float x) {
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = ::fmaf(n, -6.93145751953125e-01f, x);
    r = ::fmaf(n, -1.42860682030941723e-06f, r);
    return ::ldexpf(hw::exp2_est(r * 1.44269504088896341f), (int)n);
}
TM_HD float exp_clamped(float x) { return exp_core(::fminf(::fmaxf(x, -104.0f), 88.7f)); }
TM_HD float exp_clamped_nonpos(float x) { return exp_core(::fmaxf(x, -104.0f)); }
TM_HD float exp_nonpos(float x) { return ::expf(x); }
TM_HD float exp_fast(float x) { return ::expf(x); }
// 1/x for x in [1, 2]: hardware estimate + one cubic correction, no special cases to guard
// v_rcp_f64 is good to 4.6e-8 (tools/probes/probe_rcp_ldexp.hip): one cubic step r (1 + e + e^2),
// e = 1 - x r, brings it to ~1e-22 before rounding (three FMAs; two Newton steps need four)
TM_HD double rcp_1to2(double x) {
    const double r = hw::rcp_est(x);
    const double e = ::fma(-x, r, 1.0);
    return ::fma(::fma(e, e, e), r, r);
}
TM_HD float rcp_1to2(float x) { return hw::rcp_est(x); }
// 1/x for finite normal x > 0 (callers clamp to <= 1e300): the same, no 0 / inf / NaN guards
TM_HD double rcp_pos(double x) { return rcp_1to2(x); }
TM_HD float rcp_pos(float x) { return hw::rcp_est(x); }

#if TRAMP_ERFCX_UNIFORM
// erfcx by the Taylor recurrence about the nearest grid point, for x KNOWN to lie in [kErfcxTLo, kErfcxTHi) (the caller's
// wave vote): t_0 = f_k, t_1 = d (a f_k - 2/sqrt(pi)), t_{n+1} = (u t_n + w t_{n-1}) / (n + 1) with a = 2 x_k, u = a d,
// w = 2 d^2 (tools/gen_erfcx_ttable.py: 2e-15 on [-3, 8), 4e-14 down to -5, 6e-13 down to -7.5)
TM_HD double erfcx_tay(double x) {
    const double s = x - kErfcxTLo;
    const double kf = __builtin_rint(s * kErfcxTPer);
    const double d = ::fma(kf, -1.0 / kErfcxTPer, s);
    const double a = ::fma(kf, 2.0 / kErfcxTPer, 2.0 * kErfcxTLo);
    const double f = hw::erfcx_tval((int)kf);
    const double u = a * d, w = (d + d) * d;
    double tm = f, t = d * ::fma(a, f, -1.12837916709551257390);
    double acc = f + t;
#pragma unroll
    for (int n = 1; n < kErfcxTD; ++n) {
        const double tn = ::fma(u, t, w * tm) * (1.0 / (n + 1));
        tm = t;
        t = tn;
        acc += t;
    }
    return acc;
}
TM_HD float erfcx_tay(float x) {
    const float s = x - (float)kErfcxTLo;
    const float kf = __builtin_rintf(s * (float)kErfcxTPerF);
    const float d = ::fmaf(kf, (float)(-1.0 / kErfcxTPerF), s);
    const float a = ::fmaf(kf, (float)(2.0 / kErfcxTPerF), (float)(2.0 * kErfcxTLo));
    const float f = hw::erfcx_tval_f((int)kf);
    const float u = a * d, w = (d + d) * d;
    float tm = f, t = d * ::fmaf(a, f, -1.12837916709551257390f);
    float acc = f + t;
#pragma unroll
    for (int n = 1; n < kErfcxTDF; ++n) {
        const float tn = ::fmaf(u, t, w * tm) * (1.0f / (float)(n + 1));
        tm = t;
        t = tn;
        acc += t;
    }
    return acc;
}
// Two f32 evaluations at once on the packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth per
// issue; plain f32 issues at the f64 rate on gfx950, which is what bounds the f32 two-region channels): the two regions of
// a K7 component are the pair.  Same operations in the same order as erfcx_tay(float): bit-identical results.
typedef float tm_v2f __attribute__((ext_vector_type(2)));
#if defined(__HIPCC__)
__device__ __forceinline__ tm_v2f erfcx_tay_v2(tm_v2f x) {
    const tm_v2f s = x - (float)kErfcxTLo;
    const tm_v2f sp = s * (float)kErfcxTPerF;
    const tm_v2f kf = {__builtin_rintf(sp.x), __builtin_rintf(sp.y)};
    const tm_v2f d = __builtin_elementwise_fma(kf, (tm_v2f)(float)(-1.0 / kErfcxTPerF), s);
    const tm_v2f a = __builtin_elementwise_fma(kf, (tm_v2f)(float)(2.0 / kErfcxTPerF), (tm_v2f)(float)(2.0 * kErfcxTLo));
    const tm_v2f f = {hw::erfcx_tval_f((int)kf.x), hw::erfcx_tval_f((int)kf.y)};
    const tm_v2f u = a * d, w = (d + d) * d;
    tm_v2f tm = f, t = d * __builtin_elementwise_fma(a, f, (tm_v2f)(-1.12837916709551257390f));
    tm_v2f acc = f + t;
#pragma unroll
    for (int n = 1; n < kErfcxTDF; ++n) {
        const tm_v2f tn = __builtin_elementwise_fma(u, t, w * tm) * (1.0f / (float)(n + 1));
        tm = t;
        t = tn;
        acc += t;
    }
    return acc;
}
#endif
TM_HD void erfcx_tay_pair(float x0, float x1, float* e0, float* e1) {
#if defined(__HIP_DEVICE_COMPILE__)
    const tm_v2f e = erfcx_tay_v2(tm_v2f{x0, x1});
    *e0 = e.x;
    *e1 = e.y;
#else
    *e0 = erfcx_tay(x0);
    *e1 = erfcx_tay(x1);
#endif
}
#endif
template <bool UNI = false> TM_HD double erfcx_hot(double x) {
#if TRAMP_ERFCX_UNIFORM
    if (UNI && TM_WAVE_ALL(x >= kErfcxTLo && x < kErfcxTHi))        // a NaN fails the test: the wave takes the general path
        return erfcx_tay(x);
#endif
    const double w = ::fmin(::fabs(x), 1e300);                // erfcx(1e300) = 5.6e-301; a NaN x reaches the caller through x itself
    const double t = 4.0 * rcp_pos(4.0 + w);
    const double tk = t * (double)kErfcxK;
    const int ki = (int)tk;
    const int k = ki < kErfcxK - 1 ? ki : kErfcxK - 1;         // t in (0, 1]: tk in (0, K]
    const double s = ::fma(2.0, tk - (double)k, -1.0);
    double acc = hw::erfcx_coef(kErfcxD, k);
#pragma unroll
    for (int j = kErfcxD - 1; j >= 0; --j) acc = ::fma(acc, s, hw::erfcx_coef(j, k));
    double e = t * acc;
    if (x < 0.0) e = 2.0 * exp_clamped(x * x) - e;
    return e;
}
// f32: the same construction on the float table (4e-7 relative for x >= 0, a few ulp of float; the library's
// erfcxf is ~45 instructions)
template <bool UNI = false> TM_HD float erfcx_hot(float x) {
#if TRAMP_ERFCX_UNIFORM
    if (UNI && TM_WAVE_ALL(x >= (float)kErfcxTLo && x < (float)kErfcxTHi)) return erfcx_tay(x);
#endif
    const float w = ::fminf(::fabsf(x), 1e37f);
    const float t = 4.0f * hw::rcp_est(4.0f + w);
    const float tk = t * (float)kErfcxKF;
    const int ki = (int)tk;
    const int k = ki < kErfcxKF - 1 ? ki : kErfcxKF - 1;
    const float s = ::fmaf(2.0f, tk - (float)k, -1.0f);
    float acc = hw::erfcx_coef_f(kErfcxDF, k);
#pragma unroll
    for (int j = kErfcxDF - 1; j >= 0; --j) acc = ::fmaf(acc, s, hw::erfcx_coef_f(j, k));
    float e = t * acc;
    if (x < 0.0f) e = 2.0f * exp_clamped(x * x) - e;
    return e;
}
// Called by all threads of a block before the first erfcx_hot (Op::setup): stages the table(s) in LDS.
// UNI: also the uniform-grid table that erfcx_hot<true> reads.
template <typename T, bool UNI = false> TM_HD void erfcx_setup() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (sizeof(T) == 8) erfcx_lds_init<UNI>();
    else erfcx_lds_init_f<UNI>();
#endif
}

// the one-value table alone (the lean K7 form of the resident pipeline never leaves it: 2 KiB of LDS less)
template <typename T> TM_HD void erfcx_setup_tay() {
#if defined(__HIP_DEVICE_COMPILE__) && TRAMP_ERFCX_UNIFORM
    if (sizeof(T) == 8) {
        for (int i = threadIdx.x; i < kErfcxTK; i += blockDim.x) g_erfcx_tlds[i] = kErfcxTTable[i];
    } else {
        for (int i = threadIdx.x; i < kErfcxTKF; i += blockDim.x) g_erfcx_tlds_f[i] = kErfcxTTableF[i];
    }
    __syncthreads();
#endif
}

// largest operand handed to rcp_pos (1/x stays a normal number)
template <typename T> struct RcpCap;
template <> struct RcpCap<double> { static constexpr double value = 1e300; };
template <> struct RcpCap<float> { static constexpr float value = 1e37f; };

// tanh for the streaming kernels: sign(x) (1 - e)/(1 + e), e = exp(-2|x|) in (0, 1]; for |x| < 0.25 the
// quotient loses up to 3 bits to cancellation, where the odd series x (1 - x^2/3 + 2x^4/15 - ...) is used
// instead (the library's f64 tanh is ~170 instructions on gfx950; this is ~45).
TM_SCALAR_TABLE double kTanhSeries[8] = {-929569.0 / 638512875.0, 21844.0 / 6081075.0, -1382.0 / 155925.0, 62.0 / 2835.0,
                                         -17.0 / 315.0,          2.0 / 15.0,          -1.0 / 3.0,         1.0};     // scalar-cache operands, as kExpTaylor
TM_HD double tanh_hot(double x) {
    const double ax = ::fabs(x);
    const double e = exp_fast(-2.0 * ax);
    const double big = (1.0 - e) * rcp_1to2(1.0 + e);
    const double x2 = x * x;
    // odd Taylor series to x^15 on |x| < 0.25: truncation 1.4e-13 relative
    double p = kTanhSeries[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) p = ::fma(p, x2, kTanhSeries[i]);
    const double small = ax * p;
    const double t = ax < 0.25 ? small : big;
    return x < 0.0 ? -t : t;
}
TM_HD float tanh_hot(float x) { return ::tanhf(x); }

// tanh from the table of tanh(k / 64) (tanh_table.h) by the addition formula about the nearest grid point:
// tanh|x| = (t_k + th) / (1 + t_k th), th = tanh(|x| - x_k) by its odd series (|h| <= 1/128: the h^9 term is 3e-19).  One LDS
// read, one reciprocal of a number within 1 % of 1, ~20 VALU instructions (tanh_hot: ~50); 3e-16 against mpmath.  Past the
// table tanh rounds to 1.  A NaN stays a NaN (its index is the last entry's).  The kernel stages the table: tanh_setup().
template <typename T> TM_HD void tanh_setup() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (sizeof(T) == 8) {
        for (int i = threadIdx.x; i < kTanhK; i += blockDim.x) g_tanh_lds[i] = kTanhTable[i];
    } else {
        for (int i = threadIdx.x; i < kTanhKF; i += blockDim.x) g_tanh_lds_f[i] = kTanhTableF[i];
    }
    __syncthreads();
#endif
}
TM_HD double tanh_tab(double x) {
    const double ax = ::fabs(x);
    const double ai = ::fmin(ax, kTanhHi);                    // the index never sees a NaN ...
    const double axc = ax > kTanhHi ? kTanhHi : ax;           // ... the offset does
    const double kf = __builtin_rint(ai * kTanhPer);
    const double h = ::fma(kf, -1.0 / kTanhPer, axc);
    const double t0 = hw::tanh_tval((int)kf);
    const double h2 = h * h;
    double p = ::fma(h2, -17.0 / 315.0, 2.0 / 15.0);
    p = ::fma(p, h2, -1.0 / 3.0);
    p = ::fma(p, h2, 1.0);
    const double th = h * p;
    const double t = (t0 + th) * rcp_1to2(::fma(t0, th, 1.0));
    return ::copysign(t, x);
}
TM_HD float tanh_tab(float x) {
    const float ax = ::fabsf(x);
    const float ai = ::fminf(ax, (float)kTanhHiF);
    const float axc = ax > (float)kTanhHiF ? (float)kTanhHiF : ax;
    const float kf = __builtin_rintf(ai * (float)kTanhPerF);
    const float h = ::fmaf(kf, -1.0f / (float)kTanhPerF, axc);
    const float t0 = hw::tanh_tval_f((int)kf);
    const float h2 = h * h;
    float p = ::fmaf(h2, 2.0f / 15.0f, -1.0f / 3.0f);
    p = ::fmaf(p, h2, 1.0f);
    const float th = h * p;
    const float t = (t0 + th) * hw::rcp_est(::fmaf(t0, th, 1.0f));
    return ::copysignf(t, x);
}

// scipy.special.expit, branch-free: one exp of -|x| (never overflows), one reciprocal of 1 + e in [1, 2].
template <typename T> TM_HD T expit(T x) {
    T e = exp_clamped_nonpos(-fabs_(x));      // NaN x: see exp_clamped (the callers' b/a factor carries it)
    T r = rcp_1to2(T(1) + e);
    return x < T(0) ? e * r : r;
}

// ---- A1: message update, scalars always double (tramp/base.py:245-255) ----
TM_HD double a_new_of(double v, double a) {
    double inv = 1.0 / (v < kVFloor ? kVFloor : v);   // np.maximum propagates NaN, fmax would not
    double an = inv - a;
    // np.clip propagates NaN; fmin/fmax would swallow it
    if (an < kAMin) an = kAMin;
    if (an > kAMax) an = kAMax;
    return an;
}

// ---- A2: Gauss-Bernoulli denoiser ------------------------------------------
// tramp/priors/gauss_bernoulli_prior.py:70-77, tramp/beliefs/sparse.py:9-22,
// tramp/beliefs/normal.py:3-4.  a = ax + 1/var (per instance), lg = log(2 pi / a).
template <typename T> TM_HD RV<T> gb_posterior(T bx, T a, T lg, T b0, T eta) {
    T b = bx + b0;
    T xi = T(0.5) * (b * b / a + lg) - eta;
    T s = expit(xi);
    T ba = b / a;
    RV<T> o;
    o.r = s * ba;
    o.v = s / a + s * (T(1) - s) * (ba * ba);
    return o;
}

// Same denoiser with the per-instance scalars hoisted: inv_a = 1/a, c1 = 0.5/a,
// c0 = 0.5 log(2 pi / a) - eta.  No division left in the element loop.
template <typename T> struct GbConst { T inv_a, c1, c0, b0; };

template <typename T> TM_HD RV<T> gb_posterior_fast(T bx, const GbConst<T>& k) {
    T b = bx + k.b0;
    T xi = fma_(b * b, k.c1, k.c0);
    T s = expit(xi);
    T ba = b * k.inv_a;
    RV<T> o;
    o.r = s * ba;
    o.v = s * fma_(T(1) - s, ba * ba, k.inv_a);      // s/a + s (1-s) (b/a)^2
    return o;
}

// ---- A3: soft threshold (tramp/priors/map_L1_norm_prior.py:11-17,66-71) -----
// The support test is |b| > gamma on the same operands as the reference.
template <typename T> TM_HD bool l1_support(T bx, T gamma) { return fabs_(bx) > gamma; }

template <typename T> TM_HD RV<T> l1_posterior(T bx, T inv_ax, T gamma) {
    T shrink = fmax_(T(0), T(1) - gamma / fabs_(bx)) * bx;   // gamma/0 = inf -> 0
    RV<T> o;
    o.r = inv_ax * shrink;
    o.v = l1_support(bx, gamma) ? inv_ax : T(0);
    return o;
}

// ---- half-line truncated normal ------------------------------------------------
// tramp/utils/truncated_normal.py: truncated_normal_mean/_var/_log_proba (:234-281)
// through G0_inf/G1_inf/G2_inf (:218-231), F1_inf/F2_inf (:92-94,147-149), log_Phi
// (:21-29).  side = +1: z in [0,inf), side = -1: z in (-inf,0].
// One erfcx per region: log_Phi(-side*y0) = log(0.5*erfcx(side*y0/sqrt2)) - y0^2/2
// shares its erfcx with G1/G2.
template <typename T> struct LogPhiCut;
template <> struct LogPhiCut<double> { static constexpr double value = 30.0; }; // reference mask x < 30
// f32: erfcxf overflows below -9.2; log Phi(8) = -6e-16 is far under f32 resolution.
template <> struct LogPhiCut<float> { static constexpr float value = 8.0f; };

template <typename T> struct HalfLine { T mean, var, logp; bool mask; };

template <typename T> TM_HD HalfLine<T> halfline(T r0, T v0, int side) {
    T s0 = sqrt_(v0);
    T y0 = (T(0) - r0) / s0;
    T x = y0 / T(kSqrt2);
    T sd = T(side);
    T e = erfcx_(sd * x);
    T g1 = T(kSqrt2OverPi) * (sd / e);
    T g2 = T(kTwoOverSqrtPi) * (sd * x / e);
    HalfLine<T> h;
    h.mean = r0 + s0 * g1;
    h.var = v0 * (T(1) + g2 - g1 * g1);
    T u = -sd * y0;                         // argument of log_Phi
    h.mask = u < LogPhiCut<T>::value;
    h.logp = h.mask ? (log_(T(0.5) * e) - T(0.5) * (u * u)) : T(0);
    return h;
}

// Per-region scalars of one instance: a = az + slope^2 ax, v0 = 1/a, s0 = sqrt(v0) are formed
// once with the reference's (IEEE) operations, so that y0 = (0 - b/a)/s0 - the operand of the
// log_Phi branch mask - is bit-identical to NumPy's.  Everything downstream of y0 uses
// reciprocals (no mask depends on it).
template <typename T> struct RegionConst { T a, v0, s0, inv_s0, half_log2piv0; };

template <typename T> TM_HD RegionConst<T> region_const(double a) {
    RegionConst<T> c;
    c.a = T(a);
    c.v0 = T(1) / c.a;
    c.s0 = sqrt_(c.v0);
    c.inv_s0 = T(1) / c.s0;
    c.half_log2piv0 = T(0.5) * log_(T(2 * kPi) * c.v0);
    return c;
}

template <typename T, bool LOGP>
TM_HD HalfLine<T> halfline_fast(T b, const RegionConst<T>& c, int side, T& r0, T zb = T(0)) {
    T y0;
    if (LOGP) {
        r0 = b / c.a;                          // exact: mask operand
        y0 = (zb - r0) / c.s0;                 // exact: mask operand (zb = 0 for Sgn/Abs/Relu)
    } else {
        r0 = b * c.v0;                         // no mask downstream: reciprocals (2 x ~16 instructions saved)
        y0 = (zb - r0) * c.inv_s0;
    }
    T sd = T(side);
    T x = sd * (y0 * T(0.70710678118654752440));
    T e = erfcx_hot(x);
    T ie = rcp_pos(fmin_(e, RcpCap<T>::value));       // e = inf (x << 0) -> ~0; NaN travels through x and r0
    T g1 = T(kSqrt2OverPi) * (sd * ie);
    T g2 = T(kTwoOverSqrtPi) * (x * ie);
    HalfLine<T> h;
    h.mean = r0 + c.s0 * g1;
    h.var = c.v0 * (T(1) + g2 - g1 * g1);
    h.mask = true;
    h.logp = T(0);
    if (LOGP) {
        T u = -sd * y0;
        h.mask = u < LogPhiCut<T>::value;
        h.logp = h.mask ? (log_(T(0.5) * e) - T(0.5) * (u * u)) : T(0);
    }
    return h;
}

// Variant for the two-region merge of pwl_posterior_fast: instead of logp = log(0.5 erfcx) - u^2/2 it
// returns the FACTOR f = 0.5 erfcx (1 where the reference's log_Phi mask replaces the term by 0), its
// reciprocal, and the exponent part q = -u^2/2 (0 under the mask) - the softmax over two regions needs
// only the ratio of the two probabilities, so no logarithm is ever taken (two f64 logs, ~200
// instructions per component, saved).
template <typename T> struct HalfLineF { T mean, var, e, ie, y0; bool mask; };

template <typename T, bool UNI = true>
TM_HD HalfLineF<T> halfline_factor(T b, const RegionConst<T>& c, int side, T& r0, T zb = T(0)) {
    // The reference's mask is u = -side (zb - b/a)/s0 < cut with IEEE divisions.  The reciprocal forms
    // below differ from those by a few ulp, which can only matter within ~1e-12 of the threshold: there
    // (practically never, and then for the whole wave at once) the exact operands are formed, so the mask
    // stays bit-identical to NumPy's while the common path saves two f64 divisions per region.
    r0 = b * c.v0;
    T y0 = (zb - r0) * c.inv_s0;
    T sd = T(side);
    if (fabs_(fabs_(y0) - LogPhiCut<T>::value) < T(sizeof(T) == 4 ? 1e-5 : 1e-9) * LogPhiCut<T>::value) {
        r0 = b / c.a;
        y0 = (zb - r0) / c.s0;
    }
    T x = sd * (y0 * T(0.70710678118654752440));
    T e = erfcx_hot<UNI>(x);
    T ie = rcp_pos(fmin_(e, RcpCap<T>::value));       // e = inf (x << 0) -> ~0; NaN travels through x and r0
    T g1 = T(kSqrt2OverPi) * (sd * ie);
    T g2 = T(kTwoOverSqrtPi) * (x * ie);
    HalfLineF<T> h;
    h.mean = r0 + c.s0 * g1;
    h.var = c.v0 * (T(1) + g2 - g1 * g1);
    h.mask = -sd * y0 < LogPhiCut<T>::value;          // u = -side y0, the argument of log_Phi
    h.e = e;
    h.ie = ie;
    h.y0 = y0;
    return h;
}

// ---- A6: sign likelihood (tramp/likelihoods/sgn_likelihood.py:32-37,
//          tramp/beliefs/positive.py:12-17) -----------------------------------------
template <typename T> TM_HD RV<T> sgn_lik_posterior(T bz, T y, T az, T v0 /* 1/az */) {
    T b = bz * y;
    HalfLine<T> h = halfline<T>(b / az, v0, +1);
    RV<T> o;
    o.r = y * h.mean;
    o.v = h.var;
    return o;
}

template <typename T> TM_HD RV<T> sgn_lik_posterior_fast(T bz, T y, const RegionConst<T>& c) {
    T r0;
    HalfLine<T> h = halfline_fast<T, false>(bz * y, c, +1, r0);
    RV<T> o;
    o.r = y * h.mean;
    o.v = h.var;
    return o;
}

// Lean form for the resident pipeline (round 5; no control flow, erfcx from the one-value table).  x = -b y / sqrt(2 az):
// below the table's lower end (-7.5: the observation agrees with the message by more than 10 standard deviations) the
// truncation no longer acts - erfcx >= 5e24, the corrections s0 g1 and v0 (g2 - g1^2) are below 2e-25 of their terms
// whether erfcx is exact or the table's edge value - so the argument is simply clamped there.  ABOVE the table (x >= 8: a
// disagreement of 11 standard deviations) the tail needs erfcx accurately: `dev` > 0 asks the caller for a redo on
// sgn_lik_posterior_fast; NaN / infinite inputs make it NaN.
template <typename T> TM_HD RV<T> sgn_lik_posterior_lean(T bz, T y, const RegionConst<T>& c, T* dev) {
#if TRAMP_ERFCX_UNIFORM
    const T r0 = (bz * y) * c.v0;
    const T x = ((T(0) - r0) * c.inv_s0) * T(0.70710678118654752440);
    const T hi_v = x - fmin_(x, T(kErfcxTHi));
    *dev = fma_(hi_v, hi_v, x - x);
    const T xc = fmin_(fmax_(x, T(kErfcxTLo)), T(kErfcxTHi));
    const T e = erfcx_tay(xc);
    const T ie = rcp_pos(e);                              // e in [0.07, 5e24]
    const T g1 = T(kSqrt2OverPi) * ie;
    const T g2 = T(kTwoOverSqrtPi) * (xc * ie);
    RV<T> o;
    o.r = y * (r0 + c.s0 * g1);
    o.v = c.v0 * (T(1) + g2 - g1 * g1);
    return o;
#else
    *dev = T(0);
    return sgn_lik_posterior_fast<T>(bz, y, c);
#endif
}

#if defined(__HIPCC__) && TRAMP_ERFCX_UNIFORM
// Two f32 components at once on the packed-f32 pipe (erfcx_tay_v2): the same operations in the same order as the scalar form.
__device__ __forceinline__ void sgn_lik_posterior_lean_v2(tm_v2f bz, tm_v2f y, const RegionConst<float>& c, tm_v2f* r, tm_v2f* v, tm_v2f* dev) {
    const tm_v2f r0 = (bz * y) * c.v0;
    const tm_v2f x = ((0.0f - r0) * c.inv_s0) * 0.70710678118654752440f;
    const tm_v2f xh = {fmin_(x.x, (float)kErfcxTHi), fmin_(x.y, (float)kErfcxTHi)};
    const tm_v2f hi_v = x - xh;
    *dev = __builtin_elementwise_fma(hi_v, hi_v, x - x);
    const tm_v2f xc = {fmax_(xh.x, (float)kErfcxTLo), fmax_(xh.y, (float)kErfcxTLo)};
    const tm_v2f e = erfcx_tay_v2(xc);
    const tm_v2f ie = {rcp_pos(e.x), rcp_pos(e.y)};
    const tm_v2f g1 = (float)kSqrt2OverPi * ie;
    const tm_v2f g2 = (float)kTwoOverSqrtPi * (xc * ie);
    *r = y * (r0 + c.s0 * g1);
    *v = c.v0 * (1.0f + g2 - g1 * g1);
}
#endif

// ---- (Leaky)Relu likelihood: y = max(z, 0) + slope_neg * min(z, 0) observed
// tramp/likelihoods/piecewise_linear_likelihood.py:9-63,201-215,279-291.  Regions pos (slope 1) and
// neg (slope s): a region with non-zero slope pins z = (y - x0)/slope exactly (v = 0) when that z
// lies strictly inside it; the flat neg region of Relu (s = 0) contributes a half-line
// truncated normal when y == 0.  No region contains y -> softmax of (-inf, -inf) = NaN, as the
// reference.
template <typename T> TM_HD RV<T> relu_lik_posterior(T bz, T y, T slope_neg, const RegionConst<T>& c) {
    RV<T> o;
    const T nan = T(__builtin_nan(""));
    if (y > T(0)) {                       // pos region: z = y
        o.r = y;
        o.v = T(0);
    } else if (slope_neg != T(0)) {       // leaky: neg region contains z = y/s iff z < 0
        o.r = (y < T(0)) ? y / slope_neg : nan;
        o.v = (y < T(0)) ? T(0) : nan;
    } else if (y == T(0)) {               // relu: flat region, z in (-inf, 0]
        T r0;
        HalfLine<T> h = halfline_fast<T, false>(bz, c, -1, r0);
        o.r = h.mean;
        o.v = h.var;
    } else {
        o.r = nan;
        o.v = nan;
    }
    return o;
}

// ---- A7: abs likelihood (tramp/likelihoods/abs_likelihood.py:31-36,
//          tramp/beliefs/binary.py:8-13) ------------------------------------------
// TAB: tanh from the staged table (tanh_setup; the streaming kernels)
template <typename T, bool TAB = false> TM_HD RV<T> abs_lik_posterior(T bz, T y) {
    T t = TAB ? tanh_tab(bz * y) : tanh_hot(bz * y);
    RV<T> o;
    o.r = y * t;
    o.v = (y * y) * (T(1) - t * t);
    return o;
}

// ---- A8: piecewise-linear channels ---------------------------------------------
// tramp/channels/activation/piecewise_linear_channel.py:27-53, region tables :80-106
// (order [pos, neg]); tramp/utils/linear_region.py:32-64.
// kind: 0 = sgn, 1 = abs, 2 = relu;  k: 0 = pos region, 1 = neg region.
TM_HD constexpr double pwl_slope(int kind, int k) {
    return kind == 0 ? 0.0 : (kind == 1 ? (k == 0 ? 1.0 : -1.0) : (k == 0 ? 1.0 : 0.0));
}
TM_HD constexpr double pwl_x0(int kind, int k) {
    return kind == 0 ? (k == 0 ? 1.0 : -1.0) : 0.0;
}

struct PwlSpec { double slope[2], x0[2], zb; };

template <typename T> struct PwlOut { T r, v; unsigned mask; };

// Half line with an arbitrary boundary zb (truncated_normal_mean/_var/_log_proba with
// zmin = zb, zmax = inf resp. zmin = -inf, zmax = zb).
template <typename T> TM_HD HalfLine<T> halfline_at(T r0, T v0, int side, T zb) {
    T s0 = sqrt_(v0);
    T y0 = (zb - r0) / s0;
    T x = y0 / T(kSqrt2);
    T sd = T(side);
    T e = erfcx_(sd * x);
    T g1 = T(kSqrt2OverPi) * (sd / e);
    T g2 = T(kTwoOverSqrtPi) * (sd * x / e);
    HalfLine<T> h;
    h.mean = r0 + s0 * g1;
    h.var = v0 * (T(1) + g2 - g1 * g1);
    T u = -sd * y0;
    h.mask = u < LogPhiCut<T>::value;
    h.logp = h.mask ? (log_(T(0.5) * e) - T(0.5) * (u * u)) : T(0);
    return h;
}

// Literal restatement for an arbitrary two-half-line spec (the SE quadratures use it).
template <typename T, bool FWD>
TM_HD PwlOut<T> pwl_posterior_spec(T bz, T bx, T az, T ax, const PwlSpec& sp) {
    T rs[2], vs[2], As[2];
    unsigned mask = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const T slope = T(sp.slope[k]), x0 = T(sp.x0[k]);
        const int side = (k == 0) ? +1 : -1;
        T a = az + slope * slope * ax;
        T b = bz + slope * (bx - ax * x0);
        T r0 = b / a, v0 = T(1) / a;
        HalfLine<T> h = halfline_at<T>(r0, v0, side, T(sp.zb));
        As[k] = T(0.5) * log_(T(2 * kPi) * v0) + T(0.5) * (r0 * r0) / v0 + h.logp
                - T(0.5) * ax * (x0 * x0) + bx * x0;
        rs[k] = FWD ? slope * h.mean + x0 : h.mean;
        vs[k] = FWD ? slope * slope * h.var : h.var;
        mask |= (h.mask ? 1u : 0u) << k;
    }
    T m = fmax_(As[0], As[1]);
    T e0 = exp_(As[0] - m), e1 = exp_(As[1] - m);
    T se = e0 + e1;
    T p0 = e0 / se, p1 = e1 / se;
    PwlOut<T> o;
    o.r = p0 * rs[0] + p1 * rs[1];
    T d = rs[0] - rs[1];
    o.v = p0 * vs[0] + p1 * vs[1] + p0 * p1 * (d * d);
    o.mask = mask;
    return o;
}

// FWD: posterior of x = f(z) (a = ax side); BWD: posterior of z.
template <typename T, int KIND, bool FWD>
TM_HD PwlOut<T> pwl_posterior(T bz, T bx, T az, T ax) {
    T rs[2], vs[2], As[2];
    unsigned mask = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const T slope = T(pwl_slope(KIND, k));
        const T x0 = T(pwl_x0(KIND, k));
        const int side = (k == 0) ? +1 : -1;
        T a = az + slope * slope * ax;
        T b = bz + slope * (bx - ax * x0);
        T r0 = b / a;
        T v0 = T(1) / a;
        HalfLine<T> h = halfline<T>(r0, v0, side);
        As[k] = T(0.5) * log_(T(2 * kPi) * v0) + T(0.5) * (r0 * r0) / v0 + h.logp
                - T(0.5) * ax * (x0 * x0) + bx * x0;
        if (FWD) {
            rs[k] = slope * h.mean + x0;
            vs[k] = slope * slope * h.var;
        } else {
            rs[k] = h.mean;
            vs[k] = h.var;
        }
        mask |= (h.mask ? 1u : 0u) << k;
    }
    // scipy.special.softmax over the two regions
    T m = fmax_(As[0], As[1]);
    T e0 = exp_(As[0] - m), e1 = exp_(As[1] - m);
    T se = e0 + e1;
    T p0 = e0 / se, p1 = e1 / se;
    PwlOut<T> o;
    o.r = p0 * rs[0] + p1 * rs[1];
    T d = rs[0] - rs[1];
    o.v = p0 * vs[0] + p1 * vs[1] + p0 * p1 * (d * d);
    o.mask = mask;
    return o;
}

// Hoisted form used by the streaming kernels: RegionConst per region (a, 1/a, sqrt(1/a),
// 0.5 log(2 pi v0)) computed once per instance.
// KIND 3 = generic pair of half-line regions [zb, inf) and (-inf, zb] with run-time slopes and
// offsets: LeakyReluChannel (slopes 1 / s, zb = 0; piecewise_linear_channel.py:72-77) and
// AsymmetricAbsChannel (slopes +1 / -1, zb = shift; :94-99).
template <typename T> struct PwlConst {
    RegionConst<T> reg[2];
    T az, ax;
    T slope[2], x0[2], zb, s_ratio;
    T kx[2], gv[2];            // moment form (pwl_moment_form): x_k = kx_k b_k, G_k = side_k sqrt(2/pi) v0_k
};

template <int KIND> TM_HD PwlSpec pwl_spec_of(const PwlSpec& rt) {
    if (KIND == 3) return rt;
    PwlSpec s;
    s.slope[0] = pwl_slope(KIND, 0); s.slope[1] = pwl_slope(KIND, 1);
    s.x0[0] = pwl_x0(KIND, 0); s.x0[1] = pwl_x0(KIND, 1);
    s.zb = 0.0;
    return s;
}

template <typename T, int KIND> TM_HD PwlConst<T> pwl_const(double az, double ax, const PwlSpec& rt = PwlSpec{}) {
    PwlConst<T> c;
    c.az = T(az);
    c.ax = T(ax);
    const PwlSpec sp = pwl_spec_of<KIND>(rt);
    c.zb = T(sp.zb);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        c.slope[k] = T(sp.slope[k]);
        c.x0[k] = T(sp.x0[k]);
        c.reg[k] = region_const<T>((double)(c.az + c.slope[k] * c.slope[k] * c.ax));
    }
    c.s_ratio = c.reg[1].s0 * c.reg[0].inv_s0;        // exp(hl_1 - hl_0), hl = 0.5 log(2 pi v0)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const T sd = (k == 0) ? T(1) : T(-1);
        // x_k = side_k (0 - b v0) inv_s0 / sqrt2
        c.kx[k] = (T(0) - sd) * (c.reg[k].v0 * c.reg[k].inv_s0) * T(0.70710678118654752440);
        c.gv[k] = sd * T(kSqrt2OverPi) * c.reg[k].v0;
    }
    return c;
}

// UNI: erfcx from the uniform-grid table (22 KiB of LDS) or from the small reciprocal-mapped one (2 KiB; the resident
// pipeline, whose LDS holds the parked posterior means)
template <typename T, int KIND, bool FWD, bool UNI = true>
TM_HD PwlOut<T> pwl_posterior_regions(T bz, T bx, const PwlConst<T>& c) {
    T rs[2], vs[2], r0s[2];
    HalfLineF<T> h[2];
    unsigned mask = 0;
    const T zb = KIND < 3 ? T(0) : c.zb;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int side = (k == 0) ? +1 : -1;
        // Sgn/Abs/Relu: slopes in {0, 1, -1} and offsets in {0, 1, -1} are compile-time, and the products
        // with 0 and 1 are not formed at all (IEEE arithmetic may not fold x * 0); KIND 3: run-time table
        const double sl = pwl_slope(KIND < 3 ? KIND : 0, k), xc = pwl_x0(KIND < 3 ? KIND : 0, k);
        T b;
        if (KIND < 3) {
            const T inner = (xc == 0.0) ? bx : bx - c.ax * T(xc);
            b = (sl == 0.0) ? bz : (sl == 1.0 ? bz + inner : bz - inner);
        } else {
            b = bz + c.slope[k] * (bx - c.ax * c.x0[k]);
        }
        h[k] = halfline_factor<T, UNI>(b, c.reg[k], side, r0s[k], zb);
        if (!FWD) {
            rs[k] = h[k].mean;
            vs[k] = h[k].var;
        } else if (KIND < 3) {
            const T m = (sl == 0.0) ? T(0) : (sl == 1.0 ? h[k].mean : -h[k].mean);
            rs[k] = (xc == 0.0) ? m : (sl == 0.0 ? T(xc) : m + T(xc));
            vs[k] = (sl == 0.0) ? T(0) : h[k].var;
        } else {
            rs[k] = c.slope[k] * h[k].mean + c.x0[k];
            vs[k] = c.slope[k] * c.slope[k] * h[k].var;
        }
        mask |= (h[k].mask ? 1u : 0u) << k;
    }
    // softmax over two regions (piecewise_linear_channel.py:27-29) = logistic of A0 - A1 with the region
    // log-partitions (linear_region.py:55-64)
    //   A_k = 0.5 a_k r0_k^2 + hl_k + [log(0.5 erfcx(x_k)) - 0.5 u_k^2] + X_k,   X_k = -0.5 ax x0_k^2 + bx x0_k,
    // hl_k = 0.5 log(2 pi v0_k), u_k^2 = a_k (zb - r0_k)^2, the bracket replaced by 0 where u_k >= 30 (the
    // reference's log_Phi mask).  t = exp(A1 - A0), p0 = 1 / (1 + t), p1 = t / (1 + t): the small
    // probability is formed directly, never as 1 - big.
    T t;
    if (TM_WAVE_ALL(mask == 3u)) {
        // no lane is under the mask: 0.5 a r0^2 - 0.5 u^2 = a zb (r0 - zb / 2) in closed form (0 for zb = 0:
        // the Gaussian normaliser cancels the tail exponent), exp(hl_1 - hl_0) = s0_1 / s0_0 per instance,
        // no logarithm - and for Abs / Relu (X_k = 0) no exponential either
        const T ratio = (h[1].e * h[0].ie) * c.s_ratio;
        if (KIND == 1 || KIND == 2) {
            t = ratio;
        } else {
            T D;
            if (KIND == 0) {
                D = T(-2) * bx;                                  // X_1 - X_0 = bx (x0_1 - x0_0)
            } else {
                T Dk[2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    Dk[k] = c.reg[k].a * zb * (r0s[k] - T(0.5) * zb) - T(0.5) * c.ax * (c.x0[k] * c.x0[k]) + bx * c.x0[k];
                D = Dk[1] - Dk[0];
            }
            const T E = exp_clamped(T(0.5) * D);                 // applied in two halves: t finite whenever it can be
            t = (ratio * E) * E;
        }
    } else {
        T Qs[2], fs[2], ifs[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const T x0 = KIND < 3 ? T(pwl_x0(KIND < 3 ? KIND : 0, k)) : c.x0[k];
            const T u = h[k].y0;
            Qs[k] = c.reg[k].half_log2piv0 + T(0.5) * (r0s[k] * r0s[k]) * c.reg[k].a
                    + (h[k].mask ? T(-0.5) * (u * u) : T(0)) - T(0.5) * c.ax * (x0 * x0) + bx * x0;
            fs[k] = h[k].mask ? T(0.5) * h[k].e : T(1);
            ifs[k] = h[k].mask ? T(2) * h[k].ie : T(1);
        }
        const T E = exp_clamped(T(-0.5) * (Qs[0] - Qs[1]));
        t = ((fs[1] * ifs[0]) * E) * E;
    }
    t = fmin_(t, RcpCap<T>::value);                              // capped: p0 = 1e-300 stands for 0, p1 = t rr = 1
    const T rr = rcp_pos(T(1) + t);
    const T p0 = rr;
    const T p1 = t * rr;
    PwlOut<T> o;
    o.r = p0 * rs[0] + p1 * rs[1];
    T d = rs[0] - rs[1];
    o.v = p0 * vs[0] + p1 * vs[1] + p0 * p1 * (d * d);
    // A NaN or infinite message makes the reference's softmax NaN in every channel and direction (checked on the reference:
    // inf - inf in the region log-partitions).  Here a region of slope 0 does not look at its mean (Sgn forward: r = p0 -
    // p1), and erfcx_hot swallows a NaN argument: carry the non-finite input into r and v explicitly.
    const T poison = (bz - bz) + (bx - bx);             // 0 for finite messages
    o.r += poison;
    o.v += poison;
    o.mask = mask;
    return o;
}

// ---- K7, moment form (round 5).  Sgn / Abs / Relu: two half lines meeting at 0, slopes and offsets in {0, 1, -1}.
// With e_k = erfcx(x_k), x_k = -side_k b_k / sqrt(2 a_k), the region weights of the reference's softmax are
// W_k = s0_k e_k E_k (E_k = exp(X_k - max X), X_k = bx x0_k: 1 for Abs / Relu), and the truncated-normal moments
// (truncated_normal.py:234-266) carry 1/e_k, which the weight cancels:
//     p_k mean_k              = [W_k r0_k + G_k E_k] / S,                    G_k = side_k sqrt(2/pi) v0_k,  S = W_0 + W_1
//     p_k (var_k + (mean_k - c)^2) = [W_k (v0_k + (r0_k - c)^2) + G_k E_k (r0_k - 2c)] / S
// (the 1/e^2 terms of var_k and mean_k^2 cancel exactly: sqrt(2/pi)^2 = 2/pi).  So the posterior
//     r = c + M1,  v = M2 - M1^2,   M1 = sum_k [..]/S,  M2 = sum_k [..]/S
// costs ONE reciprocal (of S) instead of three (1/e_0, 1/e_1, 1/(1 + t)) and no per-region mean / variance at all:
// ~65 VALU instructions per component against 123-153 for the region-by-region form above.  The centre c is the r0 of the
// heavier region, so M2 - M1^2 cancels at most like (distance between the region means)^2 / v - never like r^2 / v.
// Forward direction (posterior of x = slope z + x0): a region of slope 0 is the point mass x0_k; slope +-1 enters with
// rho_k = slope_k r0_k and G_k -> slope_k G_k.
// Taken when EVERY lane of the wave is in the domain spelled out in the function (UNI: both arguments on the erfcx grid
// [-7.5, 8), or - Abs / Relu - one of them below it while the other is >= -4; else: far from the reference's
// log_Phi mask threshold, |x| < 0.98 cut / sqrt2, where e is finite), so mask = 3 by construction; any other wave - a lane near
// or under the mask, a NaN, an infinity - takes pwl_posterior_regions, the literal region-by-region form.
template <typename T> struct PwlMomentLim;
template <> struct PwlMomentLim<double> { static constexpr double value = 0.98 * 30.0 * 0.70710678118654752440; };
template <> struct PwlMomentLim<float> { static constexpr float value = 0.98f * 8.0f * 0.70710678f; };

// VOTE = false (the resident pipeline's lean form): no wave vote and no early return - every lane is evaluated with its
// erfcx arguments clamped into the domain, and the function returns whether THIS lane was inside it; the caller flags
// the instance and redoes it on the region-by-region form (ew_resident.h).
// `dev` (VOTE = false): sum of squares of what the clamps moved - 0 inside the domain, positive outside, NaN for a NaN /
// infinite message (a float the caller accumulates: lane masks kept alive across the evaluation cost the resident
// pipeline its registers).
template <typename T, int KIND, bool FWD, bool UNI, bool VOTE = true>
TM_HD bool pwl_moment_form(T bz, T bx, const PwlConst<T>& c, PwlOut<T>& o, T* dev = nullptr) {
    static_assert(KIND >= 0 && KIND < 3, "Sgn / Abs / Relu only");
    constexpr double sl0 = pwl_slope(KIND, 0), sl1 = pwl_slope(KIND, 1);
    // b_k = bz + slope_k (bx - ax x0_k): x0 = 0 wherever the slope is not (Abs, Relu); slope 0: b_k = bz
    const T b0 = (sl0 == 0.0) ? bz : bz + bx;
    const T b1 = (sl1 == 0.0) ? bz : (sl1 == 1.0 ? bz + bx : bz - bx);
    T x0 = b0 * c.kx[0], x1 = b1 * c.kx[1];
    bool ok;
    constexpr bool GRID = UNI && TRAMP_ERFCX_UNIFORM;
    // Domain on the table: both arguments below its upper end (an argument above it is a region deep in its tail whose
    // weight erfcx ~ 1 / (x sqrt pi) still counts), and not below its lower end - EXCEPT, for Abs / Relu (E_k = 1), when
    // the other argument is >= -4: erfcx(x < -7.5) >= 5e24 against erfcx(>= -4) <= 1.8e7 makes the other region's share
    // < 4e-18 whether the large weight is exact or the table's edge value, and the 1 / e terms of its own moments are
    // below 2e-25 either way.  That is the common high signal-to-noise case (one region holds all the mass).
    const T xmin = fmin_(x0, x1), xmax = fmax_(x0, x1);
    constexpr bool SAT = (KIND != 0);
    // (a saturated argument must still be above the reference's log_Phi mask threshold: mask = 3 is what this form reports)
    if (GRID) ok = xmax < T(kErfcxTHi) && (xmin >= T(kErfcxTLo) || (SAT && xmax >= T(-4) && xmin > T(0) - PwlMomentLim<T>::value));
    else ok = fmax_(fabs_(x0), fabs_(x1)) < PwlMomentLim<T>::value;
    if (KIND == 0) ok = ok && fabs_(bx) <= RcpCap<T>::value;             // bx enters through exp only: a NaN must not be clamped away
    if (VOTE) {
        if (!TM_WAVE_ALL(ok)) return false;
    } else if (GRID) {
        // no vote: the deviation from the domain as a float (0 inside; NaN for a NaN / infinite message)
        const T hi_v = xmax - fmin_(xmax, T(kErfcxTHi));                                   // > 0 above the table (its last point is THi itself)
        const T lo_v = fmax_(T(kErfcxTLo) - xmin, T(0));                                   // > 0 below it
        // > 0: the other region is not negligible, or the low argument is at the mask threshold
        const T other = SAT ? fmax_(T(-4) - xmax, T(0)) + fmax_((T(0) - PwlMomentLim<T>::value) - xmin + T(1e-3), T(0)) : T(1);
        T dv = fma_(hi_v, hi_v, lo_v * other) + ((x0 - x0) + (x1 - x1));
        if (KIND == 0) dv += bx - bx;      // bx enters through exp only
        *dev = dv;
    }
    if (GRID) {                            // (a NaN becomes the lower edge: minNum / maxNum; only reachable without the vote)
        x0 = fmin_(fmax_(x0, T(kErfcxTLo)), T(kErfcxTHi));
        x1 = fmin_(fmax_(x1, T(kErfcxTLo)), T(kErfcxTHi));
    }
    T e0, e1;
#if TRAMP_ERFCX_UNIFORM
    if constexpr (GRID) {
        if constexpr (KIND == 0) {
            // Sgn: both regions have the SAME natural parameters and opposite sides, x1 = -x0 exactly: ONE table evaluation
            // at |x0| and the reflection erfcx(-t) = 2 exp(t^2) - erfcx(t) for the other (t^2 with its rounding error:
            // exp(hi) (1 + lo)) - an exp (~20 instructions) instead of a second Taylor evaluation (~32)
            const T t = fabs_(x0);
            const T ep = erfcx_tay(t);
            const T hi = t * t, lo = fma_(t, t, T(0) - hi);
            T ex = exp_core(hi);
            ex = fma_(ex, lo, ex);
            const T en = fma_(T(2), ex, T(0) - ep);
            const bool pos0 = x0 >= T(0);
            e0 = pos0 ? ep : en;
            e1 = pos0 ? en : ep;
        } else if constexpr (sizeof(T) == 4) {
            erfcx_tay_pair(x0, x1, &e0, &e1);
        } else {
            e0 = erfcx_tay(x0);
            e1 = erfcx_tay(x1);
        }
    } else
#endif
    {
        e0 = erfcx_hot<false>(x0);
        e1 = erfcx_hot<false>(x1);
    }
    // (f32: the moments below as two-lane packed operations too - 9 % fewer instructions in isolation, no faster in the
    //  pipeline (228 -> 237 us, box-normalised) and a select between the lanes of a vector becomes a dynamically indexed
    //  private array: measured in round 5 and left out)
    const T r00 = b0 * c.reg[0].v0, r01 = b1 * c.reg[1].v0;
    T W0, W1, G0 = c.gv[0], G1 = c.gv[1];
    if (KIND == 0) {
        // sign channel: same (a, b) in both regions, X_0 - X_1 = 2 bx: the lighter side carries exp(-2 |bx|)
        const T en = exp_clamped_nonpos(T(-2) * fabs_(bx));
        const bool pos = bx >= T(0);
        const T E0 = pos ? T(1) : en, E1 = pos ? en : T(1);
        W0 = e0 * E0;                      // s0 is common: left out of the weights ...
        W1 = e1 * E1;
        G0 = (c.gv[0] * c.reg[0].inv_s0) * E0;      // ... and divided out of G_k
        G1 = (c.gv[1] * c.reg[0].inv_s0) * E1;
    } else {
        W0 = e0 * c.reg[0].s0;
        W1 = e1 * c.reg[1].s0;
    }
    const T S = W0 + W1;
    const T iS = rcp_pos(S);
    o.mask = 3u;
    if (FWD && KIND == 0) {                // x = +-1: r = p0 - p1, v = 4 p0 p1
        const T p0 = W0 * iS, p1 = W1 * iS;
        o.r = p0 - p1;
        o.v = T(4) * (p0 * p1);
        return ok;
    }
    const bool heavy0 = W0 >= W1;
    T n1, n2, cc;
    if (!FWD || KIND == 1) {
        // two truncated-normal regions: rho_k = slope_k r0_k, sigma_k = slope_k side_k (backward: slope -> 1)
        const T rho0 = r00, rho1 = (FWD && KIND == 1) ? T(0) - r01 : r01;
        const T g1 = (FWD && KIND == 1) ? T(0) - G1 : G1;
        cc = (KIND == 0) ? rho0 : (heavy0 ? rho0 : rho1);
        const T d0 = rho0 - cc, d1 = rho1 - cc;
        n1 = fma_(W0, d0, fma_(W1, d1, G0 + g1));
        const T q0 = fma_(d0, d0, c.reg[0].v0), q1 = fma_(d1, d1, c.reg[KIND == 0 ? 0 : 1].v0);
        const T t0 = fma_(T(-2), cc, rho0), t1 = fma_(T(-2), cc, rho1);
        n2 = fma_(W0, q0, fma_(W1, q1, fma_(G0, t0, g1 * t1)));
    } else {
        // Relu forward: region 0 is x = z > 0 (slope 1), region 1 the point mass x = 0
        cc = heavy0 ? r00 : T(0);
        const T d0 = r00 - cc;
        n1 = fma_(W0, d0, G0) - W1 * cc;
        const T q0 = fma_(d0, d0, c.reg[0].v0);
        const T t0 = fma_(T(-2), cc, r00);
        n2 = fma_(W0, q0, fma_(G0, t0, W1 * (cc * cc)));
    }
    const T m1 = n1 * iS;
    o.r = cc + m1;
    o.v = fma_(T(0) - m1, m1, n2 * iS);
    return ok;
}

#if defined(__HIPCC__) && TRAMP_ERFCX_UNIFORM
// The lean moment form (VOTE = false, the uniform-grid erfcx table) for TWO f32 components at once on the packed-f32 pipe:
// the lanes of every two-lane operation are the two components, so there is no horizontal sum and no cross-lane select -
// the same operations in the same order as pwl_moment_form<float, ..., false> per lane.  (Plain f32 issues at the f64
// rate on gfx950; the f32 K7 updates are instruction-bound at half f64's bytes.)
__device__ __forceinline__ tm_v2f v2_min(tm_v2f a, tm_v2f b) { return tm_v2f{fmin_(a.x, b.x), fmin_(a.y, b.y)}; }
__device__ __forceinline__ tm_v2f v2_max(tm_v2f a, tm_v2f b) { return tm_v2f{fmax_(a.x, b.x), fmax_(a.y, b.y)}; }
__device__ __forceinline__ tm_v2f v2_sel(bool cx, bool cy, tm_v2f a, tm_v2f b) { return tm_v2f{cx ? a.x : b.x, cy ? a.y : b.y}; }
template <int KIND, bool FWD>
__device__ __forceinline__ void pwl_moment_form_v2(tm_v2f bz, tm_v2f bx, const PwlConst<float>& c, tm_v2f* ro, tm_v2f* vo, tm_v2f* dev) {
    static_assert(KIND >= 0 && KIND < 3, "Sgn / Abs / Relu only");
    constexpr double sl0 = pwl_slope(KIND, 0), sl1 = pwl_slope(KIND, 1);
    constexpr float THi = (float)kErfcxTHi, TLo = (float)kErfcxTLo, Lim = PwlMomentLim<float>::value;
    const tm_v2f b0 = (sl0 == 0.0) ? bz : bz + bx;
    const tm_v2f b1 = (sl1 == 0.0) ? bz : (sl1 == 1.0 ? bz + bx : bz - bx);
    tm_v2f x0 = b0 * c.kx[0], x1 = b1 * c.kx[1];
    const tm_v2f xmin = v2_min(x0, x1), xmax = v2_max(x0, x1);
    constexpr bool SAT = (KIND != 0);
    const tm_v2f hi_v = xmax - v2_min(xmax, (tm_v2f)THi);
    const tm_v2f lo_v = v2_max(TLo - xmin, (tm_v2f)0.0f);
    const tm_v2f other = SAT ? v2_max(-4.0f - xmax, (tm_v2f)0.0f) + v2_max((0.0f - Lim) - xmin + 1e-3f, (tm_v2f)0.0f) : (tm_v2f)1.0f;
    tm_v2f dv = __builtin_elementwise_fma(hi_v, hi_v, lo_v * other) + ((x0 - x0) + (x1 - x1));
    if (KIND == 0) dv += bx - bx;
    *dev = dv;
    x0 = v2_min(v2_max(x0, (tm_v2f)TLo), (tm_v2f)THi);
    x1 = v2_min(v2_max(x1, (tm_v2f)TLo), (tm_v2f)THi);
    tm_v2f e0, e1;
    if (KIND == 0) {          // x1 = -x0: one table evaluation and the reflection (pwl_moment_form)
        const tm_v2f t = {fabs_(x0.x), fabs_(x0.y)};
        const tm_v2f ep = erfcx_tay_v2(t);
        const tm_v2f hi = t * t, lo = __builtin_elementwise_fma(t, t, 0.0f - hi);
        tm_v2f ex = {exp_core(hi.x), exp_core(hi.y)};
        ex = __builtin_elementwise_fma(ex, lo, ex);
        const tm_v2f en = __builtin_elementwise_fma((tm_v2f)2.0f, ex, 0.0f - ep);
        const bool qx = x0.x >= 0.0f, qy = x0.y >= 0.0f;
        e0 = v2_sel(qx, qy, ep, en);
        e1 = v2_sel(qx, qy, en, ep);
    } else {
        e0 = erfcx_tay_v2(x0);
        e1 = erfcx_tay_v2(x1);
    }
    const tm_v2f r00 = b0 * c.reg[0].v0, r01 = b1 * c.reg[1].v0;
    tm_v2f W0, W1, G0 = c.gv[0], G1 = c.gv[1];
    if (KIND == 0) {
        const tm_v2f ab = tm_v2f{fabs_(bx.x), fabs_(bx.y)} * -2.0f;
        const tm_v2f en = {exp_clamped_nonpos(ab.x), exp_clamped_nonpos(ab.y)};
        const bool px = bx.x >= 0.0f, py = bx.y >= 0.0f;
        const tm_v2f E0 = v2_sel(px, py, (tm_v2f)1.0f, en), E1 = v2_sel(px, py, en, (tm_v2f)1.0f);
        W0 = e0 * E0;
        W1 = e1 * E1;
        G0 = (c.gv[0] * c.reg[0].inv_s0) * E0;
        G1 = (c.gv[1] * c.reg[0].inv_s0) * E1;
    } else {
        W0 = e0 * c.reg[0].s0;
        W1 = e1 * c.reg[1].s0;
    }
    const tm_v2f S = W0 + W1;
    const tm_v2f iS = {rcp_pos(S.x), rcp_pos(S.y)};
    if (FWD && KIND == 0) {
        const tm_v2f p0 = W0 * iS, p1 = W1 * iS;
        *ro = p0 - p1;
        *vo = 4.0f * (p0 * p1);
        return;
    }
    const bool hx = W0.x >= W1.x, hy = W0.y >= W1.y;
    tm_v2f n1, n2, cc;
    if (!FWD || KIND == 1) {
        const tm_v2f rho0 = r00, rho1 = (FWD && KIND == 1) ? 0.0f - r01 : r01;
        const tm_v2f g1 = (FWD && KIND == 1) ? 0.0f - G1 : G1;
        cc = (KIND == 0) ? rho0 : v2_sel(hx, hy, rho0, rho1);
        const tm_v2f d0 = rho0 - cc, d1 = rho1 - cc;
        n1 = __builtin_elementwise_fma(W0, d0, __builtin_elementwise_fma(W1, d1, G0 + g1));
        const tm_v2f q0 = __builtin_elementwise_fma(d0, d0, (tm_v2f)c.reg[0].v0);
        const tm_v2f q1 = __builtin_elementwise_fma(d1, d1, (tm_v2f)c.reg[KIND == 0 ? 0 : 1].v0);
        const tm_v2f t0 = __builtin_elementwise_fma((tm_v2f)-2.0f, cc, rho0), t1 = __builtin_elementwise_fma((tm_v2f)-2.0f, cc, rho1);
        n2 = __builtin_elementwise_fma(W0, q0, __builtin_elementwise_fma(W1, q1, __builtin_elementwise_fma(G0, t0, g1 * t1)));
    } else {
        cc = v2_sel(hx, hy, r00, (tm_v2f)0.0f);
        const tm_v2f d0 = r00 - cc;
        n1 = __builtin_elementwise_fma(W0, d0, G0) - W1 * cc;
        const tm_v2f q0 = __builtin_elementwise_fma(d0, d0, (tm_v2f)c.reg[0].v0);
        const tm_v2f t0 = __builtin_elementwise_fma((tm_v2f)-2.0f, cc, r00);
        n2 = __builtin_elementwise_fma(W0, q0, __builtin_elementwise_fma(G0, t0, W1 * (cc * cc)));
    }
    const tm_v2f m1 = n1 * iS;
    *ro = cc + m1;
    *vo = __builtin_elementwise_fma(0.0f - m1, m1, n2 * iS);
}
#endif

template <typename T, int KIND, bool FWD, bool UNI = true>
TM_HD PwlOut<T> pwl_posterior_fast(T bz, T bx, const PwlConst<T>& c) {
    if constexpr (KIND < 3) {
        PwlOut<T> o;
        if (pwl_moment_form<T, KIND, FWD, UNI>(bz, bx, c, o)) return o;
    }
    return pwl_posterior_regions<T, KIND, FWD, UNI>(bz, bx, c);
}

// ---- finite-interval truncated normal + three-region channels ------
// tramp/utils/truncated_normal.py:13-18 (switch), :32-89 (F0), :97-144 (F1), :152-200 (F2),
// :203-215 (G0/G1/G2), :234-281.  Literal restatement, Taylor "close" branch and its quirks
// included; `branch` reports which of close / neg / pos / other was taken (an index mask).
template <typename T> struct IntervalOut { T mean, var, logp; unsigned branch; };

template <typename T> TM_HD IntervalOut<T> interval_moments(T r0, T v0, T zmin, T zmax) {
    const T s0 = sqrt_(v0);
    const T ymin = (zmin - r0) / s0, ymax = (zmax - r0) / s0;
    T x = ymin / T(kSqrt2), y = ymax / T(kSqrt2);
    if (fabs_(x) > fabs_(y)) { T t = x; x = y; y = t; }           // switch: |x| <= |y|
    const bool close = fabs_(x - y) <= T(1e-7);
    const bool neg = (x < T(0)) && (y < T(0));
    const bool pos = (x > T(0)) && (y > T(0));
    T F0, F1, F2;
    unsigned br;
    const T sqrt_pi = T(1.7724538509055159);
    if (close) {
        br = 0;
        const T e = y - x, x2 = x * x;
        F0 = (-x * e + T(1.0 / 6) * (x2 - T(2)) * (e * e) - T(1.0 / 180) * (x2 * x2 + T(2) * x2 - T(8))
              + log_(T(2) * e / sqrt_pi)) - x2;
        F1 = sqrt_pi * (x + T(0.5) * e - T(1.0 / 6) * (e * e) - T(1.0 / 12) * (e * e * e)
                        + T(1.0 / 90) * x * (x2 + T(1)) * (e * e * e * e));
        F2 = sqrt_pi * (x2 - T(0.5) + x * e - T(1.0 / 3) * (x2 - T(1)) * (e * e) - T(1.0 / 3) * x * (e * e * e)
                        + T(1.0 / 90) * (T(2) * x2 * x2 + T(3) * x2 - T(8)) * (e * e * e * e));
    } else if (neg || pos) {
        br = neg ? 1u : 2u;
        const T D = exp_(x * x - y * y);
        const T den = neg ? D * erfcx_(-y) - erfcx_(-x) : erfcx_(x) - D * erfcx_(y);
        F0 = log_(fabs_(den)) - x * x;
        F1 = (T(1) - D) / den;
        F2 = (x - D * y) / den;
    } else {
        br = 3;
        const T D = exp_(x * x - y * y);
        const T den = erf_(y) - erf_(x);
        const T ex = exp_(-(x * x));
        F0 = log_(fabs_(den));
        F1 = ex * (T(1) - D) / den;
        F2 = ex * (x - D * y) / den;
    }
    const T g0 = T(-0.6931471805599453) + F0;                   // np.log(0.5) + F0
    const T g1 = T(kSqrt2OverPi) * F1;
    const T g2 = T(kTwoOverSqrtPi) * F2;
    IntervalOut<T> o;
    o.mean = r0 + s0 * g1;
    o.var = v0 * (T(1) + g2 - g1 * g1);
    o.logp = g0;
    o.branch = br;
    return o;
}

// Regions in the reference's order [pos, mid, neg]: pos = [zhi, inf), mid = [zlo, zhi],
// neg = (-inf, zlo]  (HardTanh / HardSigmoid / SymmetricDoor,
// tramp/channels/activation/piecewise_linear_channel.py:109-132).
struct Pwl3Spec { double zlo, zhi, slope[3], x0[3]; };

// mask: bit0 / bit1 = log_Phi branch of pos / neg, bits 2-3 = branch of the mid interval.
template <typename T, bool FWD>
TM_HD PwlOut<T> pwl3_posterior(T bz, T bx, T az, T ax, const Pwl3Spec& sp) {
    T rs[3], vs[3], As[3];
    unsigned mask = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T slope = T(sp.slope[k]), x0 = T(sp.x0[k]);
        const T a = az + slope * slope * ax;
        const T b = bz + slope * (bx - ax * x0);
        const T r0 = b / a, v0 = T(1) / a;
        T rz, vz, g0;
        if (k == 1) {
            IntervalOut<T> m = interval_moments<T>(r0, v0, T(sp.zlo), T(sp.zhi));
            rz = m.mean; vz = m.var; g0 = m.logp;
            mask |= m.branch << 2;
        } else {
            HalfLine<T> h = halfline_at<T>(r0, v0, k == 0 ? +1 : -1, k == 0 ? T(sp.zhi) : T(sp.zlo));
            rz = h.mean; vz = h.var; g0 = h.logp;
            mask |= (h.mask ? 1u : 0u) << (k == 0 ? 0 : 1);
        }
        As[k] = T(0.5) * log_(T(2 * kPi) * v0) + T(0.5) * (r0 * r0) / v0 + g0
                - T(0.5) * ax * (x0 * x0) + bx * x0;
        rs[k] = FWD ? slope * rz + x0 : rz;
        vs[k] = FWD ? slope * slope * vz : vz;
    }
    const T m = fmax_(As[0], fmax_(As[1], As[2]));
    const T e0 = exp_(As[0] - m), e1 = exp_(As[1] - m), e2 = exp_(As[2] - m);
    const T se = e0 + e1 + e2;
    const T p0 = e0 / se, p1 = e1 / se, p2 = e2 / se;
    PwlOut<T> o;
    o.r = p0 * rs[0] + p1 * rs[1] + p2 * rs[2];
    const T d01 = rs[0] - rs[1], d02 = rs[0] - rs[2], d12 = rs[1] - rs[2];
    o.v = p0 * vs[0] + p1 * vs[1] + p2 * vs[2]
          + (p0 * p1 * (d01 * d01) + p0 * p2 * (d02 * d02) + p1 * p2 * (d12 * d12));
    o.mask = mask;
    return o;
}

// Three-region LIKELIHOODS (tramp/likelihoods/piecewise_linear_likelihood.py:9-63,156-205,
// 293-320): the output y is observed.  A flat region (slope 0) is compatible with y == x0 and
// then holds the truncated normal of N(bz/az, 1/az) on its interval; a sloped region pins
// z = (y - x0)/slope when that lies strictly inside it (variance 0); incompatible regions get
// log-partition -inf; softmax over the log-partitions merges (two flat regions of the door
// share x0 = +1).  No compatible region -> NaN, as the reference's softmax of all -inf.
template <typename T>
TM_HD PwlOut<T> pwl3_lik_posterior(T bz, T y, T az, const Pwl3Spec& sp) {
    T rs[3], vs[3], As[3];
    unsigned mask = 0;
    const T ninf = -T(INFINITY);
    const T r0 = bz / az, v0 = T(1) / az;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T slope = T(sp.slope[k]), x0 = T(sp.x0[k]);
        rs[k] = T(0); vs[k] = T(0); As[k] = ninf;
        if (sp.slope[k] == 0.0) {
            if (y == x0) {
                T g0;
                if (k == 1) {
                    IntervalOut<T> m = interval_moments<T>(r0, v0, T(sp.zlo), T(sp.zhi));
                    rs[k] = m.mean; vs[k] = m.var; g0 = m.logp;
                    mask |= m.branch << 2;
                } else {
                    HalfLine<T> h = halfline_at<T>(r0, v0, k == 0 ? +1 : -1, k == 0 ? T(sp.zhi) : T(sp.zlo));
                    rs[k] = h.mean; vs[k] = h.var; g0 = h.logp;
                    mask |= (h.mask ? 1u : 0u) << (k == 0 ? 0 : 1);
                }
                As[k] = T(0.5) * log_(T(2 * kPi) * v0) + T(0.5) * (r0 * r0) / v0 + g0;
            }
        } else {
            const T z = (y - x0) / slope;
            const bool in = (k == 0) ? (z > T(sp.zhi)) : (k == 1 ? (z > T(sp.zlo) && z < T(sp.zhi)) : (z < T(sp.zlo)));
            if (in) {
                rs[k] = z;
                As[k] = -T(0.5) * az * (z * z) + bz * z - log_(fabs_(slope));   // the Jacobian |slope| (the reference: log(slope))
            }
        }
    }
    const T m = fmax_(As[0], fmax_(As[1], As[2]));
    const T e0 = exp_(As[0] - m), e1 = exp_(As[1] - m), e2 = exp_(As[2] - m);   // m = -inf -> NaN
    const T se = e0 + e1 + e2;
    const T p0 = e0 / se, p1 = e1 / se, p2 = e2 / se;
    PwlOut<T> o;
    o.r = p0 * rs[0] + p1 * rs[1] + p2 * rs[2];
    const T d01 = rs[0] - rs[1], d02 = rs[0] - rs[2], d12 = rs[1] - rs[2];
    o.v = p0 * vs[0] + p1 * vs[1] + p2 * vs[2]
          + (p0 * p1 * (d01 * d01) + p0 * p2 * (d02 * d02) + p1 * p2 * (d12 * d12));
    o.mask = mask;
    return o;
}

// ---- Gaussian mixture prior (tramp/priors/gaussian_mixture_prior.py:88-97, tramp/beliefs/mixture.py:23-41) ----
// p(x) = sum_k p_k N(x | m_k, s_k): with a_k = ax + 1/s_k, b_k = bx + m_k/s_k the posterior is a mixture of N(b_k/a_k, 1/a_k)
// with weights softmax_k(eta_k + normal.A(a_k, b_k)).  Per-instance scalars hoisted: inv_a = 1/a_k, c1 = 0.5/a_k,
// c0 = eta_k + 0.5 log(2 pi / a_k).  Softmax with the polynomial exp of a non-positive argument and one reciprocal of the
// normaliser; v in the centred form sum_k s_k (1/a_k + (r_k - r)^2) - the reference's half double sum over ordered pairs,
// rearranged, all terms non-negative.  At most kGmMax components, the number known at run time: the loops are unrolled
// with static indices and predicated (registers, no scratch).
constexpr int kGmMax = 8;
template <typename T> struct GmConst { int K; T inv_a[kGmMax], c0[kGmMax], c1[kGmMax], b0[kGmMax], tb0[kGmMax]; };

// The constants of component k for a message of precision ax: xi_k = eta_k + normal.A(a_k, b_k) with eta_k = log p_k -
// normal.A(a0_k, b0_k) (gaussian_mixture_prior.py:36-39) is written as c0 + c1 bx (bx + 2 b0): the b0^2 / a0 of eta and the
// b0^2 / a_k of normal.A are taken against each other analytically (b0^2 (1/a_k - 1/a0) = -ax mean b0 / a_k), so a narrow
// component far from zero (mean^2 / var = 4e4 at mean 2, var 1e-4) does not cost f32 messages their digits.
template <typename T> TM_HD void gm_const_k(GmConst<T>& g, int k, double ax, double logp, double a0, double b0) {
    const double ad = ax + a0, inv = 1.0 / ad;
    g.inv_a[k] = T(inv);
    g.c1[k] = T(0.5 * inv);
    g.c0[k] = T(logp + 0.5 * ::log(a0 * inv) - 0.5 * ax * (b0 / a0) * b0 * inv);
    g.b0[k] = T(b0);
    g.tb0[k] = T(2.0 * b0);
}

template <typename T> TM_HD RV<T> gm_posterior(T bx, const GmConst<T>& g) {
    T e[kGmMax], rk[kGmMax];
    T m = T(-INFINITY);
#pragma unroll
    for (int k = 0; k < kGmMax; ++k) {
        if (k < g.K) {
            e[k] = fma_(bx * (bx + g.tb0[k]), g.c1[k], g.c0[k]);
            rk[k] = (bx + g.b0[k]) * g.inv_a[k];
            m = fmax_(m, e[k]);
        }
    }
    T se = T(0), sr = T(0);
#pragma unroll
    for (int k = 0; k < kGmMax; ++k) {
        if (k < g.K) {
            e[k] = exp_clamped_nonpos(e[k] - m);                   // a NaN message travels through rk
            se += e[k];
            sr = fma_(e[k], rk[k], sr);
        }
    }
    const T is = rcp_pos(se);                                      // se in [1, K]
    RV<T> o;
    o.r = sr * is;
    T sv = T(0);
#pragma unroll
    for (int k = 0; k < kGmMax; ++k) {
        if (k < g.K) {
            const T d = rk[k] - o.r;
            sv = fma_(e[k], fma_(d, d, g.inv_a[k]), sv);
        }
    }
    o.v = sv * is;
    return o;
}

// The same with the number of components a compile-time constant (2..4: what mixture priors use in practice): straight-line
// code for the streaming kernels and the resident pipeline - polynomial exp of a non-positive argument, one reciprocal of
// the normaliser in [1, K], and the variance in its centred form sum_k s_k (1/a_k + (r_k - r)^2), which is the reference's
// half double sum over pairs term for term rearranged (all terms non-negative: nothing cancels).
template <typename T, int K> TM_HD RV<T> gm_posterior_k(T bx, const GmConst<T>& g) {
    T e[K], rk[K];
    T m = T(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        e[k] = fma_(bx * (bx + g.tb0[k]), g.c1[k], g.c0[k]);
        rk[k] = (bx + g.b0[k]) * g.inv_a[k];
        m = k == 0 ? e[0] : fmax_(m, e[k]);
    }
    T se = T(0), sr = T(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        e[k] = exp_clamped_nonpos(e[k] - m);                       // a NaN message travels through rk
        se += e[k];
        sr = fma_(e[k], rk[k], sr);
    }
    const T is = rcp_pos(se);
    RV<T> o;
    o.r = sr * is;
    T sv = T(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const T d = rk[k] - o.r;
        sv = fma_(e[k], fma_(d, d, g.inv_a[k]), sv);
    }
    o.v = sv * is;
    return o;
}

// ---- modulus likelihood y = |z|, z complex (tramp/likelihoods/modulus_likelihood.py:9-20, 82-93) ----
// I(x) = I1(x) / I0(x) (the reference: scipy's ive(1, x) / ive(0, x), 1 from x = 1e9 on and for NaN).  Always in double:
// x <= 25: the two power series in t = x^2 / 4 (positive terms, no cancellation; 48 terms reach 1e-17 of the sum at 25);
// x > 25: the large-argument expansions I_nu ~ e^x / sqrt(2 pi x) sum_k e_k(nu) / x^k, e_k = e_{k-1} ((2k-1)^2 - 4 nu^2) / (8k),
// 19 terms (3e-17 at 25); the common prefactor cancels in the ratio.  Odd in x.
// reciprocals 1 / k^2 and 1 / (k (k + 1)), k = 1..48, of the two power series (read through the scalar cache: the loop
// index is wave-uniform, only the trip count depends on the lane)
TM_SCALAR_TABLE double kBesselInvKK[48] = {1.0, 0.25, 0.1111111111111111, 0.0625, 0.04, 0.027777777777777776, 0.02040816326530612, 0.015625, 0.012345679012345678, 0.01, 0.008264462809917356, 0.006944444444444444, 0.005917159763313609, 0.00510204081632653, 0.0044444444444444444, 0.00390625, 0.0034602076124567475, 0.0030864197530864196, 0.002770083102493075, 0.0025, 0.0022675736961451248, 0.002066115702479339, 0.001890359168241966, 0.001736111111111111, 0.0016, 0.0014792899408284023, 0.0013717421124828531, 0.0012755102040816326, 0.0011890606420927466, 0.0011111111111111111, 0.001040582726326743, 0.0009765625, 0.0009182736455463728, 0.0008650519031141869, 0.0008163265306122449, 0.0007716049382716049, 0.0007304601899196494, 0.0006925207756232687, 0.0006574621959237344, 0.000625, 0.000594883997620464, 0.0005668934240362812, 0.0005408328826392645, 0.0005165289256198347, 0.0004938271604938272, 0.0004725897920604915, 0.0004526935264825713, 0.00043402777777777775};
TM_SCALAR_TABLE double kBesselInvKK1[48] = {0.5, 0.16666666666666666, 0.08333333333333333, 0.05, 0.03333333333333333, 0.023809523809523808, 0.017857142857142856, 0.013888888888888888, 0.011111111111111112, 0.00909090909090909, 0.007575757575757576, 0.00641025641025641, 0.005494505494505495, 0.004761904761904762, 0.004166666666666667, 0.003676470588235294, 0.0032679738562091504, 0.0029239766081871343, 0.002631578947368421, 0.002380952380952381, 0.0021645021645021645, 0.001976284584980237, 0.0018115942028985507, 0.0016666666666666668, 0.0015384615384615385, 0.0014245014245014246, 0.0013227513227513227, 0.0012315270935960591, 0.0011494252873563218, 0.001075268817204301, 0.0010080645161290322, 0.000946969696969697, 0.00089126559714795, 0.0008403361344537816, 0.0007936507936507937, 0.0007507507507507507, 0.0007112375533428165, 0.0006747638326585695, 0.000641025641025641, 0.0006097560975609756, 0.0005807200929152149, 0.0005537098560354374, 0.0005285412262156448, 0.000505050505050505, 0.0004830917874396135, 0.00046253469010175765, 0.0004432624113475177, 0.00042517006802721087};
// terms of the power series at |x|: t^k / (k!)^2 is below 1e-17 of the sum from k = 10 + 1.6 |x| on (checked against a
// 40-digit evaluation on [0, 25], tests/test_host_math.py)
TM_HD int bessel_series_terms(double ax) {
    const int k = 10 + (int)(1.6 * ax) + 1;
    return k < 48 ? k : 48;
}
TM_HD double bessel_i1_over_i0(double x) {
    if (!(x < 1e9)) return 1.0;
    const double ax = ::fabs(x);
    double ratio;
    if (ax <= 25.0) {
        const double t = 0.25 * ax * ax;
        double t0 = 1.0, t1 = 1.0, s0 = 1.0, s1 = 1.0;
        const int kmax = bessel_series_terms(ax);
        for (int k = 0; k < kmax; ++k) {
            t0 *= t * kBesselInvKK[k];
            t1 *= t * kBesselInvKK1[k];
            s0 += t0;
            s1 += t1;
        }
        ratio = (0.5 * ax) * s1 / s0;
    } else {
        const double z = 1.0 / ax;
        double t0 = 1.0, t1 = 1.0, s0 = 1.0, s1 = 1.0;
#pragma unroll
        for (int k = 1; k <= 18; ++k) {
            const double q = (double)(2 * k - 1) * (double)(2 * k - 1);
            t0 *= z * (q * (1.0 / (8.0 * (double)k)));
            t1 *= z * ((q - 4.0) * (1.0 / (8.0 * (double)k)));
            s0 += t0;
            s1 += t1;
        }
        ratio = s1 / s0;
    }
    return x < 0.0 ? -ratio : ratio;
}

// scipy's ive(0, x) = exp(-|x|) I0(x), the same two expansions as bessel_i1_over_i0
TM_HD double bessel_i0e(double x) {
    const double ax = ::fabs(x);
    if (ax <= 25.0) {
        const double t = 0.25 * ax * ax;
        double t0 = 1.0, s0 = 1.0;
        const int kmax = bessel_series_terms(ax);
        for (int k = 0; k < kmax; ++k) {
            t0 *= t * kBesselInvKK[k];
            s0 += t0;
        }
        return s0 * ::exp(-ax);
    }
    const double z = 1.0 / ax;
    double t0 = 1.0, s0 = 1.0;
#pragma unroll
    for (int k = 1; k <= 18; ++k) {
        const double q = (double)(2 * k - 1) * (double)(2 * k - 1);
        t0 *= z * (q * (1.0 / (8.0 * (double)k)));
        s0 += t0;
    }
    return s0 / ::sqrt(2 * kPi * ax);
}

// posterior of one complex component: bz = (re, im), y = |z| observed.  rz = (bz / |bz|) y I(|bz| y) (0 where bz = 0),
// v = y^2 (1 - I^2) / 2 (the 1/2 averages the two real coordinates).  |bz| is np.absolute of the complex number (hypot).
struct ModOut { double re, im, v; };
TM_HD ModOut modulus_posterior(double re, double im, double y) {
    const double b = ::hypot(re, im);
    const double I = bessel_i1_over_i0(b * y);
    const double sc = y * I;
    ModOut o;
    o.re = (b != 0.0) ? (re / b) * sc : 0.0;
    o.im = (b != 0.0) ? (im / b) * sc : 0.0;
    o.v = 0.5 * (y * y) * (1.0 - I * I);
    return o;
}

// Gaussian CDF / PDF (tramp/utils/misc.py:50-57) for the SE quadratures.
TM_HD double norm_cdf(double x) { return 0.5 * (1.0 + ::erf(x / kSqrt2)); }
TM_HD double norm_pdf(double x) { return ::exp(-0.5 * x * x) / 2.5066282746310002; }

}  // namespace tmath
