// prior_hint.h - "the consumer of the message I am writing is a separable prior".
//
// An isotropic prior update needs v = mean_i v_i(a, b_i) over the message (a, b) it receives before
// it can write a single component of its own message (tramp/base.py:250-255), which costs a whole
// extra read of b (pass 1 of ew_kernels.hip).  The kernel that WRITES b holds every b_i in a
// register: when the host announces the consumer (tramp_prior_hint), the producer's epilogue
// evaluates v_i as it stores b_i and leaves per-block partial sums, and the prior update becomes
// ONE streaming pass (read b, write b_new: the 2T algorithmic minimum of SURVEY section 8d).
// Producers that implement it: the dense channel's fold / rotated-basis epilogues (gemm_mfma.hip,
// linear.hip) and the inverse FFT's fused message update (fft_pow2.hip).
#pragma once
#include "common.h"
#include "tramp_math.h"

// The same holds for a separable LIKELIHOOD at the other end of the chain: its input message bz is written by the
// channel's forward update, its observation y is static, so the producer can evaluate v_i(az, bz_i, y_i) as it
// stores bz_i (one extra read of y_i) and the likelihood update is finalize + ONE pass that recomputes r_i.
enum { PRIOR_HINT_NONE = 0, PRIOR_HINT_GB = 1, PRIOR_HINT_L1 = 2, PRIOR_HINT_SGN_LIK = 3, PRIOR_HINT_ABS_LIK = 4 };

// device-passable part (by value in the producer's epilogue descriptor)
struct PriorHintDev {
    int kind;
    double p0, p1, p2;        // GB: a0 = 1/var, b0 = mean/var, eta;  L1: gamma
    const double* a_msg;      // [batch] precision of the message being written (final, after damping)
    double* partials;         // [batch, nblk] sum of v_i per producer block
    int nblk;                 // producer blocks per instance
    const void* y;            // likelihood kinds: the observation [batch, n], element type of the message
    size_t n;                 //                   row length of y
};

// Host side: if the context holds a hint for the message buffer `out_buf` ([batch, len]) that the
// calling kernel is about to write with `nblk` blocks per instance, reserve the partials, mark the
// hint as honoured and fill `dev`.  Returns false (dev->kind = 0) otherwise.
bool prior_hint_arm(tramp_ctx* ctx, const void* out_buf, size_t len, size_t batch, int nblk, PriorHintDev* dev);

// per-thread constants of one instance
template <typename T> struct PriorHintEval {
    int kind;
    tmath::GbConst<T> k;
    tmath::RegionConst<T> c;
    T inv_ax, gamma;
    const T* y;
    // called by ALL threads of the block (the sign likelihood stages the erfcx table in LDS: a barrier inside)
    __device__ __forceinline__ void setup(const PriorHintDev& h, size_t b) {
        kind = h.kind;
        const double ax = h.a_msg[b];
        y = reinterpret_cast<const T*>(h.y) + b * h.n;
        if (kind == PRIOR_HINT_SGN_LIK) {
            c = tmath::region_const<T>(ax);
            tmath::erfcx_setup<T>();
        } else if (kind == PRIOR_HINT_ABS_LIK) {
        } else if (kind == PRIOR_HINT_GB) {
            const double ad = ax + h.p0;
            const double inv = 1.0 / ad;
            k.inv_a = T(inv);
            k.c1 = T(0.5 * inv);
            k.c0 = T(0.5 * log(2 * tmath::kPi / ad) - h.p2);
            k.b0 = T(h.p1);
        } else {
            inv_ax = T(1.0 / ax);
            gamma = T(h.p0);
        }
    }
    // v_i of the consumer's posterior for the message component b at position idx of the instance
    // (NaN-poisoning as ew_pass1)
    __device__ __forceinline__ double v(T b, size_t idx) const {
        tmath::RV<T> o;
        if (kind == PRIOR_HINT_GB) o = tmath::gb_posterior_fast<T>(b, k);
        else if (kind == PRIOR_HINT_L1) o = tmath::l1_posterior<T>(b, inv_ax, gamma);
        else if (kind == PRIOR_HINT_SGN_LIK) o = tmath::sgn_lik_posterior_fast<T>(b, y[idx], c);
        else o = tmath::abs_lik_posterior<T>(b, y[idx]);
        double acc = (double)o.v;
        if ((o.r - o.r) != T(0)) acc += __longlong_as_double(0x7ff8000000000000LL);
        return acc;
    }
};
