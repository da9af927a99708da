// Host build of tramp_amd/csrc/tramp_math.h: a UNIT-TEST HARNESS for the per-component
// arithmetic (tests/test_host_math.py compares it with the oracle on the CPU).  It is test
// infrastructure only - nothing in tramp_amd/ loads it, and it is not a CPU fallback.
// The special functions of the streaming kernels (table erfcx, polynomial exp, tanh, the corrected reciprocal) are the
// DEVICE expressions; only the primitives of tmath::hw differ (hardware estimates, where a table is read from) and the
// library erfcx of the literal forms is a stand-in (exp(x^2) erfc(x): agreement ~1e-12 there, not bitwise).
#include <cstddef>
#include <cstdint>

#include "../../tramp_amd/csrc/tramp_math.h"

using namespace tmath;

template <typename T>
static void gb(size_t n, const T* bx, double ax, double rho, double mean, double var, T* r, T* v, int fast) {
    const double a0 = 1.0 / var, b0 = mean / var;
    const double eta = 0.5 * (b0 * b0 / a0 + log(2 * kPi / a0)) - log(rho / (1 - rho));
    const double ad = ax + a0;
    GbConst<T> k{T(1.0 / ad), T(0.5 / ad), T(0.5 * log(2 * kPi / ad) - eta), T(b0)};
    for (size_t i = 0; i < n; ++i) {
        RV<T> o = fast ? gb_posterior_fast<T>(bx[i], k)
                       : gb_posterior<T>(bx[i], T(ad), T(log(2 * kPi / ad)), T(b0), T(eta));
        r[i] = o.r;
        v[i] = o.v;
    }
}

template <typename T, int KIND, bool FWD>
static void pwl(size_t n, const T* bz, const T* bx, double az, double ax, T* r, T* v, uint8_t* mask, int fast) {
    PwlConst<T> c = pwl_const<T, KIND>(az, ax);
    for (size_t i = 0; i < n; ++i) {
        PwlOut<T> o = fast ? pwl_posterior_fast<T, KIND, FWD>(bz[i], bx[i], c)
                           : pwl_posterior<T, KIND, FWD>(bz[i], bx[i], T(az), T(ax));
        r[i] = o.r;
        v[i] = o.v;
        mask[i] = (uint8_t)o.mask;
    }
}

// fast: 0 literal region form, 1 pwl_posterior_fast on the uniform erfcx grid (UNI), 2 pwl_posterior_fast with the small table
// (UNI = false: the form the resident pipeline instantiates)
template <typename T, int KIND, bool FWD>
static void pwl_any(size_t n, const T* bz, const T* bx, double az, double ax, T* r, T* v, uint8_t* mask, int fast) {
    PwlConst<T> c = pwl_const<T, KIND>(az, ax);
    for (size_t i = 0; i < n; ++i) {
        PwlOut<T> o = fast == 0 ? pwl_posterior<T, KIND, FWD>(bz[i], bx[i], T(az), T(ax))
                                : (fast == 1 ? pwl_posterior_fast<T, KIND, FWD, true>(bz[i], bx[i], c)
                                             : pwl_posterior_fast<T, KIND, FWD, false>(bz[i], bx[i], c));
        r[i] = o.r; v[i] = o.v; mask[i] = (uint8_t)o.mask;
    }
}
template <typename T>
static void gm(size_t n, const T* bx, double ax, int K, const double* probs, const double* means, const double* vars, T* r, T* v, int fast) {
    GmConst<T> g;
    g.K = K;
    for (int k = 0; k < K; ++k) {
        gm_const_k<T>(g, k, ax, log(probs[k]), 1.0 / vars[k], means[k] / vars[k]);
    }
    for (size_t i = 0; i < n; ++i) {
        RV<T> o = (fast && K == 2) ? gm_posterior_k<T, 2>(bx[i], g)
                : (fast && K == 3) ? gm_posterior_k<T, 3>(bx[i], g)
                : (fast && K == 4) ? gm_posterior_k<T, 4>(bx[i], g) : gm_posterior<T>(bx[i], g);
        r[i] = o.r; v[i] = o.v;
    }
}
extern "C" {
void hm_gb_f64(size_t n, const double* bx, double ax, double rho, double mean, double var, double* r, double* v, int fast) { gb<double>(n, bx, ax, rho, mean, var, r, v, fast); }
void hm_gb_f32(size_t n, const float* bx, double ax, double rho, double mean, double var, float* r, float* v, int fast) { gb<float>(n, bx, ax, rho, mean, var, r, v, fast); }

void hm_l1_f64(size_t n, const double* bx, double ax, double gamma, double* r, double* v, uint8_t* mask) {
    for (size_t i = 0; i < n; ++i) {
        RV<double> o = l1_posterior<double>(bx[i], 1.0 / ax, gamma);
        r[i] = o.r; v[i] = o.v; mask[i] = l1_support<double>(bx[i], gamma);
    }
}
void hm_sgn_lik_f64(size_t n, const double* bz, const double* y, double az, double* r, double* v, int fast) {
    RegionConst<double> c = region_const<double>(az);
    for (size_t i = 0; i < n; ++i) {
        RV<double> o = fast ? sgn_lik_posterior_fast<double>(bz[i], y[i], c)
                            : sgn_lik_posterior<double>(bz[i], y[i], az, 1.0 / az);
        r[i] = o.r; v[i] = o.v;
    }
}
void hm_abs_lik_f64(size_t n, const double* bz, const double* y, double* r, double* v) {
    for (size_t i = 0; i < n; ++i) {
        RV<double> o = abs_lik_posterior<double>(bz[i], y[i]);
        r[i] = o.r; v[i] = o.v;
    }
}
void hm_pwl_f64(int kind, int fwd, size_t n, const double* bz, const double* bx, double az, double ax, double* r, double* v, uint8_t* mask, int fast) {
    switch (kind * 2 + (fwd ? 1 : 0)) {
        case 0: pwl<double, 0, false>(n, bz, bx, az, ax, r, v, mask, fast); break;
        case 1: pwl<double, 0, true>(n, bz, bx, az, ax, r, v, mask, fast); break;
        case 2: pwl<double, 1, false>(n, bz, bx, az, ax, r, v, mask, fast); break;
        case 3: pwl<double, 1, true>(n, bz, bx, az, ax, r, v, mask, fast); break;
        case 4: pwl<double, 2, false>(n, bz, bx, az, ax, r, v, mask, fast); break;
        case 5: pwl<double, 2, true>(n, bz, bx, az, ax, r, v, mask, fast); break;
    }
}
#define HM_PWL_ANY(NAME, T)                                                                                                   \
    void NAME(int kind, int fwd, size_t n, const T* bz, const T* bx, double az, double ax, T* r, T* v, uint8_t* mask, int fast) { \
        switch (kind * 2 + (fwd ? 1 : 0)) {                                                                                   \
            case 0: pwl_any<T, 0, false>(n, bz, bx, az, ax, r, v, mask, fast); break;                                        \
            case 1: pwl_any<T, 0, true>(n, bz, bx, az, ax, r, v, mask, fast); break;                                         \
            case 2: pwl_any<T, 1, false>(n, bz, bx, az, ax, r, v, mask, fast); break;                                        \
            case 3: pwl_any<T, 1, true>(n, bz, bx, az, ax, r, v, mask, fast); break;                                         \
            case 4: pwl_any<T, 2, false>(n, bz, bx, az, ax, r, v, mask, fast); break;                                        \
            case 5: pwl_any<T, 2, true>(n, bz, bx, az, ax, r, v, mask, fast); break;                                         \
        }                                                                                                                     \
    }
HM_PWL_ANY(hm_pwl_any_f64, double)
HM_PWL_ANY(hm_pwl_any_f32, float)
void hm_pwl3_f64(int fwd, size_t n, const double* bz, const double* bx, double az, double ax, double zlo, double zhi,
                 const double* slopes3, const double* x03, double* r, double* v, uint8_t* mask) {
    Pwl3Spec sp{zlo, zhi, {slopes3[0], slopes3[1], slopes3[2]}, {x03[0], x03[1], x03[2]}};
    for (size_t i = 0; i < n; ++i) {
        PwlOut<double> o = fwd ? pwl3_posterior<double, true>(bz[i], bx[i], az, ax, sp)
                               : pwl3_posterior<double, false>(bz[i], bx[i], az, ax, sp);
        r[i] = o.r; v[i] = o.v; mask[i] = (uint8_t)o.mask;
    }
}
void hm_pwl2_f64(int fwd, size_t n, const double* bz, const double* bx, double az, double ax, double sp_, double sn,
                 double xp, double xn, double zc, double* r, double* v) {
    PwlSpec sp{{sp_, sn}, {xp, xn}, zc};
    PwlConst<double> c = pwl_const<double, 3>(az, ax, sp);
    for (size_t i = 0; i < n; ++i) {
        PwlOut<double> o = fwd ? pwl_posterior_fast<double, 3, true>(bz[i], bx[i], c)
                               : pwl_posterior_fast<double, 3, false>(bz[i], bx[i], c);
        r[i] = o.r; v[i] = o.v;
    }
}
void hm_relu_lik_f64(size_t n, const double* bz, const double* y, double az, double slope, double* r, double* v) {
    RegionConst<double> c = region_const<double>(az);
    for (size_t i = 0; i < n; ++i) {
        RV<double> o = relu_lik_posterior<double>(bz[i], y[i], slope, c);
        r[i] = o.r; v[i] = o.v;
    }
}
double hm_a_new(double v, double a) { return a_new_of(v, a); }
// which: 0 erfcx_hot<false>, 1 erfcx_hot<true> (uniform-grid table), 2 exp_fast, 3 exp_clamped, 4 tanh_hot, 5 rcp_pos, 6 expit,
// 7 tanh_tab (the table form of the binary prior)
void hm_special_f64(int which, size_t n, const double* x, double* out) {
    for (size_t i = 0; i < n; ++i) {
        switch (which) {
            case 0: out[i] = erfcx_hot<false>(x[i]); break;
            case 1: out[i] = erfcx_hot<true>(x[i]); break;
            case 2: out[i] = exp_fast(x[i]); break;
            case 3: out[i] = exp_clamped(x[i]); break;
            case 4: out[i] = tanh_hot(x[i]); break;
            case 5: out[i] = rcp_pos(x[i]); break;
            case 7: out[i] = tanh_tab(x[i]); break;
            default: out[i] = expit<double>(x[i]); break;
        }
    }
}
void hm_special_f32(int which, size_t n, const float* x, float* out) {
    for (size_t i = 0; i < n; ++i) {
        switch (which) {
            case 0: out[i] = erfcx_hot<false>(x[i]); break;
            case 3: out[i] = exp_clamped(x[i]); break;
            case 7: out[i] = tanh_tab(x[i]); break;
            default: out[i] = expit<float>(x[i]); break;
        }
    }
}

// round 6: the rest of the separable priors and the modulus likelihood
// half-line prior p(x) ~ 1[x >= 0] exp(-a0 x^2 / 2 + b0 x): Exponential (a0 = 0, b0 = -1/mean), Positive (a0 = 1, b0 = 0)
void hm_halfline_prior_f64(size_t n, const double* bx, double ax, double a0, double b0, double* r, double* v) {
    RegionConst<double> c = region_const<double>(ax + a0);
    for (size_t i = 0; i < n; ++i) {
        RV<double> o = sgn_lik_posterior_fast<double>(bx[i] + b0, 1.0, c);
        r[i] = o.r; v[i] = o.v;
    }
}
void hm_gm_f64(size_t n, const double* bx, double ax, int K, const double* probs, const double* means, const double* vars, double* r, double* v, int fast) { gm<double>(n, bx, ax, K, probs, means, vars, r, v, fast); }
void hm_gm_f32(size_t n, const float* bx, double ax, int K, const double* probs, const double* means, const double* vars, float* r, float* v, int fast) { gm<float>(n, bx, ax, K, probs, means, vars, r, v, fast); }
void hm_bessel_ratio(size_t n, const double* x, double* out) {
    for (size_t i = 0; i < n; ++i) out[i] = bessel_i1_over_i0(x[i]);
}
void hm_modulus(size_t n, const double* re, const double* im, const double* y, double* rre, double* rim, double* v) {
    for (size_t i = 0; i < n; ++i) {
        ModOut o = modulus_posterior(re[i], im[i], y[i]);
        rre[i] = o.re; rim[i] = o.im; v[i] = o.v;
    }
}
}
