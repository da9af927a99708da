// ew_ops.h - the componentwise factors as functors (K1-K7 and the piecewise-linear family).
// Each Op: stage() brings the erfcx table(s) into LDS (all threads of the block; ends in a barrier, no-op for the ops
// without erfcx), prepare(b) forms the per-instance scalars (pure: no memory but the precision arrays), setup(b) =
// stage() + prepare(b); eval(in0, in1) -> (r, v, mask); a_msg / TARGET select the message being updated.
// Shared by ew_kernels.hip (tile kernels), ew_resident.h (resident single-pass pipeline) and tools/probes.
#pragma once
#include "common.h"
#include "tramp_math.h"

using tmath::RV;

// ---------------------------------------------------------------- functors
// Each Op: setup(b) loads the per-instance scalars; eval(in0, in1) -> (r, v, mask);
// a_msg / TARGET select the message being updated.
// dev: lean ops only (OpLean) - 0 where the evaluation was inside its form's domain, positive or NaN where not
template <typename T> struct EwOut { T r, v; unsigned mask; T dev = T(0); };

template <typename T> struct GbOp {       // K1; in0 = bx
    static constexpr bool HEAVY = false;
    static constexpr int NIN = 1;
    static constexpr int TARGET = 0;
    const double* ax;
    double a0, b0, eta;
    tmath::GbConst<T> k;
    double a_msg;
    __device__ static void stage() {}
    __device__ void prepare(size_t b) {
        a_msg = ax[b];
        const double ad = a_msg + a0;
        const double inv = 1.0 / ad;
        k.inv_a = T(inv);
        k.c1 = T(0.5 * inv);
        k.c0 = T(0.5 * log(2 * tmath::kPi / ad) - eta);
        k.b0 = T(b0);
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T) const {
        RV<T> o = tmath::gb_posterior_fast<T>(x0, k);
        return {o.r, o.v, 0u};
    }
};

template <typename T> struct L1Op {       // K2; in0 = bx
    static constexpr bool HEAVY = false;
    static constexpr int NIN = 1;
    static constexpr int TARGET = 0;
    const double* ax;
    double gamma;
    T inv_ax, gT;
    double a_msg;
    __device__ static void stage() {}
    __device__ void prepare(size_t b) {
        a_msg = ax[b];
        inv_ax = T(1.0 / a_msg);
        gT = T(gamma);
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T) const {
        RV<T> o = tmath::l1_posterior<T>(x0, inv_ax, gT);
        return {o.r, o.v, tmath::l1_support<T>(x0, gT) ? 1u : 0u};
    }
};

template <typename T> struct BinaryOp {   // binary prior p(x) = p+ delta(x - 1) + p- delta(x + 1); in0 = bx
    static constexpr bool HEAVY = false;
    static constexpr int NIN = 1;
    static constexpr int TARGET = 0;
    const double* ax;
    double b0;                              // natural parameter 0.5 log(p+ / p-), binary_prior.py:27
    double a_msg;
    __device__ static void stage() { tmath::tanh_setup<T>(); }
    __device__ void prepare(size_t b) { a_msg = ax[b]; }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T) const {
        const T t = tmath::tanh_tab(x0 + T(b0));                 // beliefs/binary.py:8-13
        return {t, T(1) - t * t, 0u};
    }
};

// Half-line prior p(x) ~ 1[x >= 0] exp(-a0 x^2 / 2 + b0 x): ExponentialPrior (a0 = 0, b0 = -1/mean;
// tramp/priors/exponential_prior.py:25-26, 54-60) and PositivePrior (a0 = 1, b0 = 0; positive_prior.py:23-25, 56-63).
// The posterior is positive.r / positive.v (beliefs/positive.py:12-17) at (ax + a0, bx + b0): the half-line truncated
// normal of the sign likelihood with y = +1.  in0 = bx
// LEAN (f64): the resident pipeline's form as for the sign likelihood (tmath::sgn_lik_posterior_lean: no control flow,
// one-value erfcx table; an argument beyond the table comes back in EwOut::dev and the caller redoes the instance)
template <typename T, bool LEAN = false> struct HalfLinePriorOp {
    static constexpr bool HEAVY = false;     // as the sign likelihood: recomputed in pass 2
    static constexpr int NIN = 1;
    static constexpr int TARGET = 0;
    const double* ax;
    double a0, b0;
    // The arithmetic is double whatever the element type: the variance v0 (1 + g2 - g1^2) cancels to eps x^2 at
    // x = -(b/a) sqrt(a/2), and an exponential prior sits at x = 1 / (mean sqrt(2 ax)) - 22 at ax = 1e-3, where f32
    // would leave 3 % (tests/test_gpu_r6_factors.py); f32 messages are widened on load and rounded on store.
    tmath::RegionConst<double> c;
    double a_msg;
    __device__ static void stage() {
        if (LEAN) tmath::erfcx_setup_tay<double>();
        else tmath::erfcx_setup<double>();
    }
    __device__ void prepare(size_t b) {
        a_msg = ax[b];
        c = tmath::region_const<double>(a_msg + a0);
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T) const {
        if constexpr (LEAN) {
            double dev;
            RV<double> o = tmath::sgn_lik_posterior_lean<double>((double)x0 + b0, 1.0, c, &dev);
            return {T(o.r), T(o.v), 0u, T(dev)};
        } else {
            RV<double> o = tmath::sgn_lik_posterior_fast<double>((double)x0 + b0, 1.0, c);
            return {T(o.r), T(o.v), 0u};
        }
    }
};

// GaussianMixturePrior (tramp/priors/gaussian_mixture_prior.py:36-39, 88-97; beliefs/mixture.py:23-41), at most
// tmath::kGmMax components; a0 = 1/var_k, b0 = mean_k/var_k, logp = log p_k.  in0 = bx
// KC: the number of components at compile time (2..4: straight-line code, resident pipeline), 0 = run time (up to kGmMax)
template <typename T, int KC = 0> struct GmOp {
    static constexpr bool HEAVY = false;
    static constexpr int NIN = 1;
    static constexpr int TARGET = 0;
    const double* ax;
    int K;
    double a0[tmath::kGmMax], b0[tmath::kGmMax], logp[tmath::kGmMax];
    tmath::GmConst<T> g;
    double a_msg;
    __device__ static void stage() {}
    __device__ void prepare(size_t b) {
        a_msg = ax[b];
        g.K = K;
#pragma unroll
        for (int k = 0; k < (KC ? KC : tmath::kGmMax); ++k) {
            if (KC || k < K) tmath::gm_const_k<T>(g, k, a_msg, logp[k], a0[k], b0[k]);
        }
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T) const {
        RV<T> o;
        if constexpr (KC != 0) o = tmath::gm_posterior_k<T, KC>(x0, g);
        else o = tmath::gm_posterior<T>(x0, g);
        return {o.r, o.v, 0u};
    }
};

// LEAN: the resident pipeline's form (tmath::sgn_lik_posterior_lean; a disagreement beyond the erfcx table or a non-finite
// input comes back in EwOut::dev and the caller redoes the instance on the full form)
template <typename T, bool LEAN = false> struct SgnLikOp {   // K4; in0 = bz, in1 = y
    static constexpr bool HEAVY = false;     // recomputed in pass 2 (5T moved, not 6T): f32 252 vs 309 us, f64 580 vs 616 us at 64 x 1e6
    static constexpr int NIN = 2;
    static constexpr int TARGET = 0;
    const double* az;
    tmath::RegionConst<T> c;
    double a_msg;
    __device__ static void stage() {
        if (LEAN) tmath::erfcx_setup_tay<T>();
        else tmath::erfcx_setup<T>();
    }
    __device__ void prepare(size_t b) {
        a_msg = az[b];
        c = tmath::region_const<T>(a_msg);
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T x1) const {
        if constexpr (LEAN) {
            T dev;
            RV<T> o = tmath::sgn_lik_posterior_lean<T>(x0, x1, c, &dev);
            return {o.r, o.v, 0u, dev};
        } else {
            RV<T> o = tmath::sgn_lik_posterior_fast<T>(x0, x1, c);
            return {o.r, o.v, 0u};
        }
    }
    // two components at once (f32 lean form: packed-f32 arithmetic, OpPair)
    __device__ void eval2(float a0, float a1, float y0, float y1, EwOut<float>& o0, EwOut<float>& o1) const {
        tmath::tm_v2f r, v, dev;
        tmath::sgn_lik_posterior_lean_v2(tmath::tm_v2f{a0, a1}, tmath::tm_v2f{y0, y1}, c, &r, &v, &dev);
        o0 = {r.x, v.x, 0u, dev.x};
        o1 = {r.y, v.y, 0u, dev.y};
    }
};

// K4 with the observation packed one bit per component (y in {-1, +1} is static: 2T + 1/64 T moved instead of 3T);
// only the resident pipeline reads it (ew_resident.h, ResArgs::bits).  eval's second argument is y rebuilt from its bit.
template <typename T, bool LEAN = false> struct SgnLikBitsOp {   // in0 = bz
    static constexpr bool HEAVY = false;
    static constexpr int NIN = 1;
    static constexpr int TARGET = 0;
    const double* az;
    tmath::RegionConst<T> c;
    double a_msg;
    __device__ static void stage() {
        if (LEAN) tmath::erfcx_setup_tay<T>();
        else tmath::erfcx_setup<T>();
    }
    __device__ void prepare(size_t b) {
        a_msg = az[b];
        c = tmath::region_const<T>(a_msg);
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T y) const {
        if constexpr (LEAN) {
            T dev;
            RV<T> o = tmath::sgn_lik_posterior_lean<T>(x0, y, c, &dev);
            return {o.r, o.v, 0u, dev};
        } else {
            RV<T> o = tmath::sgn_lik_posterior_fast<T>(x0, y, c);
            return {o.r, o.v, 0u};
        }
    }
    __device__ void eval2(float a0, float a1, float y0, float y1, EwOut<float>& o0, EwOut<float>& o1) const {
        tmath::tm_v2f r, v, dev;
        tmath::sgn_lik_posterior_lean_v2(tmath::tm_v2f{a0, a1}, tmath::tm_v2f{y0, y1}, c, &r, &v, &dev);
        o0 = {r.x, v.x, 0u, dev.x};
        o1 = {r.y, v.y, 0u, dev.y};
    }
};

template <typename T> struct AbsLikOp {   // K5; in0 = bz, in1 = y
    static constexpr bool HEAVY = false;      // ~35 instructions per component (tanh_tab): recomputed in pass 2 (5T moved, not 6T)
    static constexpr int NIN = 2;
    static constexpr int TARGET = 0;
    const double* az;
    double a_msg;
    __device__ static void stage() { tmath::tanh_setup<T>(); }
    __device__ void prepare(size_t b) { a_msg = az[b]; }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T x1) const {
        RV<T> o = tmath::abs_lik_posterior<T, true>(x0, x1);
        return {o.r, o.v, 0u};
    }
};

// LEAN (slope_neg == 0 only): the resident pipeline's form - no control flow; the flat region's half-line by
// tmath::sgn_lik_posterior_lean (z <= 0 is the sign likelihood's y = -1), the pinned region by selects; an observation that
// no region holds (y < 0, NaN) or a message beyond the erfcx table comes back in EwOut::dev and the caller redoes the instance.
template <typename T, bool LEAN = false> struct ReluLikOp {  // in0 = bz, in1 = y
    static constexpr bool HEAVY = true;
    static constexpr int NIN = 2;
    static constexpr int TARGET = 0;
    const double* az;
    double slope_neg;
    tmath::RegionConst<T> c;
    T sT;
    double a_msg;
    __device__ static void stage() {
        if (LEAN) tmath::erfcx_setup_tay<T>();
        else tmath::erfcx_setup<T>();
    }
    __device__ void prepare(size_t b) {
        a_msg = az[b];
        c = tmath::region_const<T>(a_msg);
        sT = T(slope_neg);
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T x1) const {
        if constexpr (LEAN) {
            T dev;
            const RV<T> h = tmath::sgn_lik_posterior_lean<T>(x0, T(-1), c, &dev);
            const bool pos = x1 > T(0), none = !(x1 >= T(0));
            dev = pos ? T(0) : dev;                                   // a pinned component does not look at the message
            dev = none ? T(__builtin_nan("")) : dev;
            return {pos ? x1 : h.r, pos ? T(0) : h.v, 0u, dev};
        } else {
            RV<T> o = tmath::relu_lik_posterior<T>(x0, x1, sT, c);
            return {o.r, o.v, 0u};
        }
    }
    __device__ void eval2(float a0, float a1, float y0, float y1, EwOut<float>& o0, EwOut<float>& o1) const {
        tmath::tm_v2f r, v, dev;
        tmath::sgn_lik_posterior_lean_v2(tmath::tm_v2f{a0, a1}, tmath::tm_v2f{-1.0f, -1.0f}, c, &r, &v, &dev);
        const bool p0 = y0 > 0.0f, p1 = y1 > 0.0f, n0 = !(y0 >= 0.0f), n1 = !(y1 >= 0.0f);
        const float nan = __builtin_nanf("");
        o0 = {p0 ? y0 : r.x, p0 ? 0.0f : v.x, 0u, n0 ? nan : (p0 ? 0.0f : dev.x)};
        o1 = {p1 ? y1 : r.y, p1 ? 0.0f : v.y, 0u, n1 ? nan : (p1 ? 0.0f : dev.y)};
    }
};

template <typename T> struct Pwl3LikOp {  // three-region likelihoods; in0 = bz, in1 = y
    static constexpr bool HEAVY = true;
    static constexpr int NIN = 2;
    static constexpr int TARGET = 0;
    const double* az;
    tmath::Pwl3Spec spec;
    T azT;
    double a_msg;
    __device__ static void stage() {}
    __device__ void prepare(size_t b) {
        a_msg = az[b];
        azT = T(a_msg);
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T x1) const {
        tmath::PwlOut<T> o = tmath::pwl3_lik_posterior<T>(x0, x1, azT, spec);
        return {o.r, o.v, o.mask};
    }
};

template <typename T, bool FWD> struct Pwl3Op {   // three-region channels; in0 = bz, in1 = bx
    static constexpr bool HEAVY = true;
    static constexpr int NIN = 2;
    static constexpr int TARGET = FWD ? 1 : 0;
    const double* az;
    const double* ax;
    tmath::Pwl3Spec spec;
    T azT, axT;
    double a_msg;
    __device__ static void stage() {}
    __device__ void prepare(size_t b) {
        const double z = az[b], x = ax[b];
        azT = T(z);
        axT = T(x);
        a_msg = FWD ? x : z;
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T x1) const {
        tmath::PwlOut<T> o = tmath::pwl3_posterior<T, FWD>(x0, x1, azT, axT, spec);
        return {o.r, o.v, o.mask};
    }
};

// LEAN (the resident pipeline): the moment form only; a wave with a lane outside its domain still evaluates it and
// reports the instance through `eval_ok` = false, for the caller to redo on the literal form (ew_resident.h: redo flags)
template <typename T, int KIND, bool FWD, bool UNI = true, bool LEAN = false> struct PwlOp {   // K7; in0 = bz, in1 = bx
    static constexpr bool HEAVY = true;
    static constexpr int NIN = 2;
    static constexpr int TARGET = FWD ? 1 : 0;
    const double* az;
    const double* ax;
    tmath::PwlSpec spec;          // used by KIND 3 only
    tmath::PwlConst<T> c;
    double a_msg;
    __device__ static void stage() {
        if (LEAN) tmath::erfcx_setup_tay<T>();
        else tmath::erfcx_setup<T, UNI>();
    }
    __device__ void prepare(size_t b) {
        const double z = az[b], x = ax[b];
        c = tmath::pwl_const<T, KIND>(z, x, spec);
        a_msg = FWD ? x : z;
    }
    __device__ void setup(size_t b) { stage(); prepare(b); }
    __device__ EwOut<T> eval(T x0, T x1) const {
        if constexpr (LEAN && KIND < 3) {
            tmath::PwlOut<T> o;
            T dev;
            tmath::pwl_moment_form<T, KIND, FWD, UNI, false>(x0, x1, c, o, &dev);
            return {o.r, o.v, o.mask, dev};
        } else {
            tmath::PwlOut<T> o = tmath::pwl_posterior_fast<T, KIND, FWD, UNI>(x0, x1, c);
            return {o.r, o.v, o.mask};
        }
    }
    // two components at once (f32 lean form on the packed-f32 pipe, OpPair)
    __device__ void eval2(float z0, float z1, float xa, float xb, EwOut<float>& o0, EwOut<float>& o1) const {
        if constexpr (KIND < 3 && sizeof(T) == 4) {
            tmath::tm_v2f r, v, dev;
            tmath::pwl_moment_form_v2<KIND, FWD>(tmath::tm_v2f{z0, z1}, tmath::tm_v2f{xa, xb}, c, &r, &v, &dev);
            o0 = {r.x, v.x, 3u, dev.x};
            o1 = {r.y, v.y, 3u, dev.y};
        }
    }
};

// Only the L1 prior reports a support count (MAP_L1NormPrior's mask, map_L1_norm_prior.py:16-17): the other
// ops do not carry the 64-bit counter through their loops.
template <typename Op> struct OpBits { static constexpr bool value = false; };
template <typename T, bool LEAN> struct OpBits<SgnLikBitsOp<T, LEAN>> { static constexpr bool value = true; };

// Lean ops (the resident pipeline's K7): eval() covers the moment form's domain only and reports a lane outside it in
// EwOut::dev (non-zero or NaN); the pipeline poisons the instance's a_new and the caller redoes it on the full form
// (ew_resident.h, ew_kernels.hip: pwl_dir).
template <typename Op> struct OpLean { static constexpr bool value = false; };
template <typename T, int KIND, bool FWD, bool UNI> struct OpLean<PwlOp<T, KIND, FWD, UNI, true>> { static constexpr bool value = true; };
template <typename T> struct OpLean<SgnLikOp<T, true>> { static constexpr bool value = true; };
template <typename T> struct OpLean<SgnLikBitsOp<T, true>> { static constexpr bool value = true; };
template <typename T> struct OpLean<ReluLikOp<T, true>> { static constexpr bool value = true; };
template <typename T> struct OpLean<HalfLinePriorOp<T, true>> { static constexpr bool value = true; };

// OpPair: f32 lean ops with eval2 (two components per call on the packed-f32 pipe)
template <typename Op> struct OpPair { static constexpr bool value = false; };
template <> struct OpPair<SgnLikOp<float, true>> { static constexpr bool value = true; };
template <> struct OpPair<SgnLikBitsOp<float, true>> { static constexpr bool value = true; };
template <> struct OpPair<ReluLikOp<float, true>> { static constexpr bool value = true; };
template <int KIND, bool FWD, bool UNI> struct OpPair<PwlOp<float, KIND, FWD, UNI, true>> { static constexpr bool value = KIND < 3; };

template <typename Op> struct OpCounts { static constexpr bool value = false; };
template <typename T> struct OpCounts<L1Op<T>> { static constexpr bool value = true; };

// ---------------------------------------------------------------- pass 1: moments
template <typename T> __device__ __forceinline__ bool not_finite(T x) { return (x - x) != T(0); }

// Where an update leaves its results (tramp/base.py:250-255): the message (a_new, b_new), optionally damped against
// (a_old, b_old), and on request the posterior (r, v), the L1 support count and the branch masks.
struct MsgOut {
    double damping;
    const double* a_old;
    const void* b_old;
    double* a_new;
    void* b_new;
    void* r;
    double* v;
    unsigned long long* count;
    uint8_t* mask;
};

// native 16-byte vector of the element type (the nontemporal builtins take these, not double2 / float4)
template <typename T> struct NativeVec;
template <> struct NativeVec<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct NativeVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
