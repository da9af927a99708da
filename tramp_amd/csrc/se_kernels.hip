// se_kernels.hip - state-evolution errors (K13, K14): Gaussian averages of the factor
// posterior variances, evaluated by brute-force composite Gauss-Legendre quadrature.
//
// The reference integrates with adaptive QUADPACK through Python callbacks
// (tramp/utils/integration.py:13-73: quad / dblquad on [-10, 10] resp. [-10, 10]^2,
// ~4e4 scalar posterior evaluations and 4-7 s per 2-D call).  On the GPU the same
// integrals are embarrassingly parallel over nodes:
//   1-D: 2048 panels x 8-point GL, panels graded cubically towards the point where the
//        integrand's argument b crosses 0 (the denoisers change regime on a width
//        ~ 1/sqrt(SNR) around it; a uniform mesh under-resolves that at large a).
//   2-D: tensor grid of 8-point GL panels (default 768 panels = 6144 nodes per axis,
//        3.8e7 evaluations per call), one block per strip of x1 nodes.
// VALU / transcendental bound, no meaningful memory traffic: report evaluations/s.
// Inputs and outputs are HOST arrays of `count` independent parameter points; partial
// sums are folded in a fixed order (deterministic).
#include <vector>

#include "common.h"
#include "tramp_math.h"

using tmath::norm_cdf;
using tmath::norm_pdf;

__constant__ double kGLx[8] = {-0.9602898564975363, -0.7966664774136267, -0.5255324099163290,
                               -0.1834346424956498, 0.1834346424956498,  0.5255324099163290,
                               0.7966664774136267,  0.9602898564975363};
__constant__ double kGLw[8] = {0.1012285362903763, 0.2223810344533745, 0.3137066458778873,
                               0.3626837833783620, 0.3626837833783620, 0.3137066458778873,
                               0.2223810344533745, 0.1012285362903763};

constexpr int kPanels1D = 2048;
static unsigned long long g_se_evals = 0;      // integrand evaluations of the last K13 / K14 call (bench.py)
static int g_panels2d = 768;      // per axis; tunable through TRAMP_SE_PANELS2D

// Node `idx` (0 .. 8*P-1) of a mesh on [-10, 10] with P/2 panels on each side of xc, graded
// as xc + L t^3, t uniform in [0,1]; returns x and the weight dx (Jacobian included).
__device__ __forceinline__ void graded_node(int idx, int P, double xc, double& x, double& w) {
    const int half = P / 2;
    const int panel = idx >> 3, g = idx & 7;
    const bool right = panel >= half;
    const int pi = right ? panel - half : panel;
    const double L = right ? (10.0 - xc) : (xc + 10.0);
    const double h = 0.5 / half;
    const double t = (pi + 0.5) * (1.0 / half) + h * kGLx[g];
    const double t2 = t * t;
    x = right ? xc + L * t2 * t : xc - L * t2 * t;
    w = L * 3.0 * t2 * h * kGLw[g];
}

// The same grading on a segment [lo, hi] around xc in it: P/2 panels on each side.
__device__ __forceinline__ void graded_node_seg(int idx, int P, double lo, double hi, double xc, double& x, double& w) {
    const int half = P / 2;
    const int panel = idx >> 3, g = idx & 7;
    const bool right = panel >= half;
    const int pi = right ? panel - half : panel;
    const double L = right ? (hi - xc) : (xc - lo);
    const double h = 0.5 / half;
    const double t = (pi + 0.5) * (1.0 / half) + h * kGLx[g];
    const double t2 = t * t;
    x = right ? xc + L * t2 * t : xc - L * t2 * t;
    w = L * 3.0 * t2 * h * kGLw[g];
}

// Observations that leave z continuous (a half line, an interval, their unions) - the information they carry about a z that the
// incoming message already pins to 1 / az is the small quantity  delta = 1 - az v;  v itself then only says 1 / az and
// a_new = 1 / v - az cancels (state evolution from an informed start runs at az = 1e11, delta ~ 1e-6).  For a truncated normal on
// pieces [alpha_i, beta_i] (standardised about the message's mean):  p (1 - az var) = A + B^2 / Z  with
//     A = sum_i beta_i phi(beta_i) - alpha_i phi(alpha_i),  B = sum_i phi(alpha_i) - phi(beta_i),  Z = sum_i Phi(beta_i) - Phi(alpha_i)
// - every term is non-zero only within a few 1 / sqrt(az) of a boundary, so delta is summed without cancellation and
// v = (1 - delta) / az.  One piece of one observation:
struct GainAcc {
    double A = 0.0, B = 0.0, Z = 0.0;
    __device__ __forceinline__ void piece(double alpha, double beta) {      // +-INFINITY allowed
        const double pa = isinf(alpha) ? 0.0 : norm_pdf(alpha), pb = isinf(beta) ? 0.0 : norm_pdf(beta);
        A += (isinf(beta) ? 0.0 : beta * pb) - (isinf(alpha) ? 0.0 : alpha * pa);
        B += pa - pb;
        Z += (isinf(beta) ? 1.0 : norm_cdf(beta)) - (isinf(alpha) ? 0.0 : norm_cdf(alpha));
    }
    __device__ __forceinline__ double gain() const { return Z > 0.0 ? A + B * B / Z : 0.0; }
};

// ---------------------------------------------------------------- K13: GB prior, sgn likelihood
// mode 0: Gauss-Bernoulli prior  (gauss_bernoulli_prior.py:59-62,116-122; base_prior.py:66-69)
//         v = (1-rho) E_{N(0, ax)}[f] + rho E_{N(ax mean, ax + ax^2 var)}[f],  f(b) = v(ax, b)
// mode 1: sign likelihood        (sgn_likelihood.py:26-27,79-92; base_likelihood.py:78-81)
//         v = sum_{y=+-1} E_{N(0, mz + mz^2 tau)}[ p(az, y b) v(az, y b) ]
struct Se1dParams {
    const double* p0;   // ax | az
    const double* p1;   // rho | tau_z
    const double* p2;   // mean
    const double* p3;   // var
    tmath::Pwl3Spec spec3;   // mode 3 (three-region likelihood) only
    struct { int K; double probs[tmath::kGmMax], means[tmath::kGmMax], vars[tmath::kGmMax]; } gm;   // mode 5 only
};

__global__ __launch_bounds__(kBlock) void se_1d_kernel(int mode, Se1dParams P, double* __restrict__ out) {
    __shared__ double smem[kBlock / kWave];
    const int pt = blockIdx.x;
    double total = 0.0;
    bool gain_mode = false;
    if (mode == 0) {
        const double ax = P.p0[pt], rho = P.p1[pt], mean = P.p2[pt], var = P.p3[pt];
        const double a0 = 1.0 / var, b0 = mean / var;
        const double eta = 0.5 * (b0 * b0 / a0 + log(2 * tmath::kPi / a0)) - log(rho / (1 - rho));
        const double a = ax + a0, lg = log(2 * tmath::kPi / a);
        for (int br = 0; br < 2; ++br) {
            const double m = br ? ax * mean : 0.0;
            const double s = br ? sqrt(ax + ax * ax * var) : sqrt(ax);
            const double wt = br ? rho : 1.0 - rho;
            // the denoiser switches regime where b = bx + b0 crosses 0
            double xc = (s > 0) ? (-b0 - m) / s : 0.0;
            xc = fmin(fmax(xc, -10.0), 10.0);
            double acc = 0.0;
            for (int i = threadIdx.x; i < 8 * kPanels1D; i += kBlock) {
                double x, w;
                graded_node(i, kPanels1D, xc, x, w);
                tmath::RV<double> o = tmath::gb_posterior<double>(m + s * x, a, lg, b0, eta);
                acc += w * norm_pdf(x) * o.v;
            }
            total += wt * acc;
        }
    } else if (mode == 1) {
        const double az = P.p0[pt], tau = P.p1[pt];
        const double mz = az - 1.0 / tau;
        const double s = sqrt(mz + mz * mz * tau);
        const double v0 = 1.0 / az;
        double acc = 0.0;
        for (int i = threadIdx.x; i < 8 * kPanels1D; i += kBlock) {
            double x, w;
            graded_node(i, kPanels1D, 0.0, x, w);
            const double bz = s * x;
            double f = 0.0;
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                const double y = yy ? -1.0 : 1.0;
                const double r0 = (y * bz) / az;
                GainAcc g;                                       // y z > 0: the half line (0, inf) about y r0
                g.piece((0 - r0) / sqrt(v0), INFINITY);
                f += g.gain();
            }
            acc += w * norm_pdf(x) * f;
        }
        gain_mode = true;
        total = acc;
    }
    if (mode == 4) {
        // binary prior (binary_prior.py:82-86, base_prior.py:66-69):
        //   v = p+ E_{N(+ax, ax)}[f] + p- E_{N(-ax, ax)}[f],  f(b) = 1 - tanh(b + b0)^2
        const double ax = P.p0[pt], p_pos = P.p1[pt];
        const double b0 = 0.5 * log(p_pos / (1.0 - p_pos));
        const double s = sqrt(ax);
        for (int br = 0; br < 2; ++br) {
            const double m = br ? -ax : ax;
            const double wt = br ? 1.0 - p_pos : p_pos;
            double xc = (s > 0) ? (-b0 - m) / s : 0.0;            // where b + b0 crosses 0
            xc = fmin(fmax(xc, -10.0), 10.0);
            double acc = 0.0;
            for (int i = threadIdx.x; i < 8 * kPanels1D; i += kBlock) {
                double x, w;
                graded_node(i, kPanels1D, xc, x, w);
                const double t = tanh(m + s * x + b0);
                acc += w * norm_pdf(x) * (1.0 - t * t);
            }
            total += wt * acc;
        }
    }
    if (mode == 5) {
        // Gaussian mixture prior (gaussian_mixture_prior.py:132-137 through base_prior.py:66-69):
        //   v = sum_k p_k E_{N(ax m_k, ax + ax^2 s_k)}[f],  f(b) = v(ax, b) of the mixture posterior (beliefs/mixture.py:34-41)
        const double ax = P.p0[pt];
        const int K = P.gm.K;
        tmath::GmConst<double> g;
        g.K = K;
#pragma unroll
        for (int k = 0; k < tmath::kGmMax; ++k) {
            if (k < K) {
                tmath::gm_const_k<double>(g, k, ax, log(P.gm.probs[k]), 1.0 / P.gm.vars[k], P.gm.means[k] / P.gm.vars[k]);
            }
        }
        for (int br = 0; br < K; ++br) {
            const double m = ax * P.gm.means[br];
            const double s = sqrt(ax + ax * ax * P.gm.vars[br]);
            // the mesh is graded towards the centre of the mixture in b (the posterior changes component around it)
            double bc = 0.0;
            for (int k = 0; k < K; ++k) bc += P.gm.probs[k] * P.gm.means[k];
            double xc = (s > 0) ? (ax * bc - m) / s : 0.0;
            xc = fmin(fmax(xc, -10.0), 10.0);
            double acc = 0.0;
            for (int i = threadIdx.x; i < 8 * kPanels1D; i += kBlock) {
                double x, w;
                graded_node(i, kPanels1D, xc, x, w);
                tmath::RV<double> o = tmath::gm_posterior<double>(m + s * x, g);
                acc += w * norm_pdf(x) * o.v;
            }
            total += P.gm.probs[br] * acc;
        }
    }
    if (mode == 2) {
        // ReluLikelihood: only the flat region has a non-zero variance
        // (piecewise_linear_likelihood.py:124-131): E_{N(0, mz + mz^2 tau)}[ p(z <= 0) v_neg ]
        const double az = P.p0[pt], tau = P.p1[pt];
        const double mz = az - 1.0 / tau;
        const double s = sqrt(mz + mz * mz * tau);
        const double v0 = 1.0 / az;
        double acc = 0.0;
        for (int i = threadIdx.x; i < 8 * kPanels1D; i += kBlock) {
            double x, w;
            graded_node(i, kPanels1D, 0.0, x, w);
            const double r0 = (s * x) / az;
            GainAcc g;                                           // y = 0: z <= 0; y > 0 pins z (its whole mass is gain)
            g.piece(-INFINITY, (0 - r0) / sqrt(v0));
            acc += w * norm_pdf(x) * (g.gain() + (1.0 - g.Z));
        }
        gain_mode = true;
        total = acc;
    }
    if (mode == 3) {
        // three-region likelihood (piecewise_linear_likelihood.py:120-146, 210-214): a sloped region
        // pins z (variance 0); each FLAT region R adds E_{bz ~ N(0, mz + mz^2 tau)}[ P_R(bz) v(bz, y = x0_R) ]
        // with v the merged posterior variance at that observation.
        const double az = P.p0[pt], tau = P.p1[pt];
        const double mz = az - 1.0 / tau;
        const double s = sqrt(mz + mz * mz * tau);
        const double v0 = 1.0 / az, s0 = sqrt(v0);
        const tmath::Pwl3Spec sp = P.spec3;
        // the integrand lives at the two region boundaries (r0 = zlo, zhi): one mesh segment per boundary, graded at it
        const double xlo = fmin(fmax(sp.zlo * az / s, -9.999), 9.999), xhi = fmin(fmax(sp.zhi * az / s, -9.999), 9.999);
        const double xmid = 0.5 * (xlo + xhi);
        double acc = 0.0;
        for (int seg = 0; seg < 2; ++seg) {
            for (int i = threadIdx.x; i < 8 * kPanels1D; i += kBlock) {
                double x, w;
                if (seg == 0) graded_node_seg(i, kPanels1D, -10.0, xmid, xlo, x, w);
                else graded_node_seg(i, kPanels1D, xmid, 10.0, xhi, x, w);
                const double r0 = (s * x) / az;
                const double a1 = (sp.zlo - r0) / s0, a2 = (sp.zhi - r0) / s0;       // boundaries, standardised
                // regions [pos, mid, neg] = (zhi, inf), (zlo, zhi), (-inf, zlo); flat ones with the same x0 are ONE observation
                double f = 0.0;
                bool done[3] = {false, false, false};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double lo_k = (k == 0) ? a2 : (k == 1 ? a1 : -INFINITY), hi_k = (k == 0) ? INFINITY : (k == 1 ? a2 : a1);
                    if (sp.slope[k] != 0.0) {                    // a sloped region pins z: its mass is gain
                        f += (isinf(hi_k) ? 1.0 : norm_cdf(hi_k)) - (isinf(lo_k) ? 0.0 : norm_cdf(lo_k));
                        continue;
                    }
                    if (done[k]) continue;
                    GainAcc g;
#pragma unroll
                    for (int j = k; j < 3; ++j) {
                        if (sp.slope[j] != 0.0 || sp.x0[j] != sp.x0[k] || done[j]) continue;
                        g.piece((j == 0) ? a2 : (j == 1 ? a1 : -INFINITY), (j == 0) ? INFINITY : (j == 1 ? a2 : a1));
                        done[j] = true;
                    }
                    f += g.gain();
                }
                acc += w * norm_pdf(x) * f;
            }
        }
        gain_mode = true;
        total = acc;
    }
    total = block_sum(total, smem);
    // the modes that summed delta = 1 - az v (sign, ReLU, three-region likelihoods) hand out v = (1 - delta) / az
    if (threadIdx.x == 0) out[pt] = gain_mode ? (1.0 - total) / P.p0[pt] : total;
}

// ---------------------------------------------------------------- K14: 2-D averages
// kind 0/1/2: Sgn/Abs/Relu channel, dir 0 fwd / 1 bwd
//   (channels/base_channel.py:29-41, piecewise_linear_channel.py:60-64, linear_region.py:78-101)
// kind 3: abs likelihood (abs_likelihood.py:25-26,60-69)
struct Se2dParams {
    const double* az;
    const double* ax;
    const double* tau;
    tmath::PwlSpec spec;     // KIND 4 (generic two-half-line channel) only
    tmath::Pwl3Spec spec3;   // KIND 5 (three-region channel) only
};

template <int KIND, bool FWD>
__device__ __forceinline__ double pwl_integrand(double x1, double x2, double az, double ax, double tau,
                                                const tmath::PwlSpec& spec) {
    const double u_eff = fmax(0.0, az * tau - 1.0);
    double total = 0.0;
    const tmath::PwlSpec sp = tmath::pwl_spec_of<(KIND == 4 ? 3 : KIND)>(spec);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double slope = sp.slope[k], x0 = sp.x0[k];
        const int side = k == 0 ? +1 : -1;
        const double mean_x = ax * x0;
        const double sx2 = ax * (slope * slope * ax * tau + 1.0);
        double bz, bx;
        if (ax == 0.0 || u_eff == 0.0 || slope == 0.0) {
            bz = sqrt(az * u_eff) * x1;                          // gaussian_measure_2d
            bx = mean_x + sqrt(sx2) * x2;
        } else {
            const double c00 = az * u_eff, c01 = slope * ax * u_eff;   // gaussian_measure_2d_full
            const double l00 = sqrt(c00), l10 = c01 / l00, l11 = sqrt(sx2 - l10 * l10);
            bz = l00 * x1;
            bx = l10 * x1 + l11 * x2 + mean_x;
        }
        // LinearRegion.proba_ab (linear_region.py:78-81)
        const double a = az + slope * slope * ax;
        const double b = bz + slope * (bx - ax * x0);
        const double y0 = (sp.zb - b / a) / sqrt(1.0 / a);
        const double p = side > 0 ? 1.0 - norm_cdf(y0) : norm_cdf(y0);
        tmath::PwlOut<double> o = tmath::pwl_posterior_spec<double, FWD>(bz, bx, az, ax, sp);
        total += p * o.v;
    }
    return total;
}

// Three-region channels: sum over [pos, mid, neg] of E[ p_k(bz, bx) v(bz, bx) ]
template <bool FWD>
__device__ __forceinline__ double pwl3_integrand(double x1, double x2, double az, double ax, double tau,
                                                 const tmath::Pwl3Spec& sp) {
    const double u_eff = fmax(0.0, az * tau - 1.0);
    double total = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double slope = sp.slope[k], x0 = sp.x0[k];
        const double mean_x = ax * x0;
        const double sx2 = ax * (slope * slope * ax * tau + 1.0);
        double bz, bx;
        if (ax == 0.0 || u_eff == 0.0 || slope == 0.0) {
            bz = sqrt(az * u_eff) * x1;
            bx = mean_x + sqrt(sx2) * x2;
        } else {
            const double c00 = az * u_eff, c01 = slope * ax * u_eff;
            const double l00 = sqrt(c00), l10 = c01 / l00, l11 = sqrt(sx2 - l10 * l10);
            bz = l00 * x1;
            bx = l10 * x1 + l11 * x2 + mean_x;
        }
        const double a = az + slope * slope * ax;
        const double b = bz + slope * (bx - ax * x0);
        const double r0 = b / a, s0 = sqrt(1.0 / a);
        double p;                                              // truncated_normal_proba
        if (k == 0) p = 1.0 - norm_cdf((sp.zhi - r0) / s0);
        else if (k == 2) p = norm_cdf((sp.zlo - r0) / s0);
        else p = norm_cdf((sp.zhi - r0) / s0) - norm_cdf((sp.zlo - r0) / s0);
        tmath::PwlOut<double> o = tmath::pwl3_posterior<double, FWD>(bz, bx, az, ax, sp);
        total += p * o.v;
    }
    return total;
}

__device__ __forceinline__ double abs_lik_integrand(double x1, double x2, double az, double tau) {
    const double mz = az - 1.0 / tau;
    const double z = sqrt(tau) * x1;
    const double bz = mz * z + sqrt(mz) * x2;
    const double y = fabs(z);
    const double t = tanh(bz * y);
    return (y * y) * (1.0 - t * t);
}

// KIND 6: ModulusLikelihood.beliefs_measure for u_eff = az tau - 1 > 0 (modulus_likelihood.py:111-120):
//   b = sqrt(az u_eff) x1, y = b / az + x2 / sqrt(az),
//   f = (2 pi / sqrt(u_eff)) relu(b) relu(y) ive(0, b y) v(b, y),  v = y^2 (1 - I(b y)^2) / 2   (:64-70)
__device__ __forceinline__ double modulus_lik_integrand(double x1, double x2, double az, double tau) {
    const double u_eff = az * tau - 1.0;
    const double b = sqrt(az * u_eff) * x1;
    const double y = b / az + x2 / sqrt(az);
    if (!(b > 0.0) || !(y > 0.0)) return 0.0;
    const double I = tmath::bessel_i1_over_i0(b * y);
    return (2 * tmath::kPi / sqrt(u_eff)) * b * y * tmath::bessel_i0e(b * y) * (0.5 * (y * y) * (1.0 - I * I));
}

// grid: (strips, count); each block integrates a strip of x1 nodes against ALL x2 nodes.
// nodes: [2][n] = x, w*pdf(x) for the uniform mesh (precomputed by se_nodes_kernel).
template <int KIND, bool FWD>
__global__ __launch_bounds__(kBlock) void se_2d_kernel(Se2dParams P, const double* __restrict__ nodes, int n,
                                                       int strip, double* __restrict__ partials) {
    __shared__ double smem[kBlock / kWave];
    const int pt = blockIdx.y;
    const double az = P.az[pt], ax = (KIND == 3 || KIND == 6) ? 0.0 : P.ax[pt], tau = P.tau[pt];
    const double* xs = nodes;
    const double* ws = nodes + n;
    const int i0 = blockIdx.x * strip;
    const int i1 = min(n, i0 + strip);
    double acc = 0.0;
    for (int i = i0; i < i1; ++i) {
        const double x1 = xs[i], w1 = ws[i];
        double row = 0.0;
        for (int j = threadIdx.x; j < n; j += kBlock) {
            const double x2 = xs[j];
            double f;
            if (KIND == 3) f = abs_lik_integrand(x1, x2, az, tau);
            else if (KIND == 6) f = modulus_lik_integrand(x1, x2, az, tau);
            else if (KIND == 5) f = pwl3_integrand<FWD>(x1, x2, az, ax, tau, P.spec3);
            else f = pwl_integrand<(KIND == 3 || KIND == 5 || KIND == 6 ? 0 : KIND), FWD>(x1, x2, az, ax, tau, P.spec);
            row += ws[j] * f;
        }
        acc += w1 * row;
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) partials[(size_t)pt * gridDim.x + blockIdx.x] = acc;
}

__global__ void se_nodes_kernel(int panels, double* __restrict__ nodes) {
    const int n = panels * 8;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double h = 10.0 / panels;                 // half panel width
    const double c = -10.0 + (2 * (i >> 3) + 1) * h;
    const double x = c + h * kGLx[i & 7];
    nodes[i] = x;
    nodes[n + i] = h * kGLw[i & 7] * norm_pdf(x);
}

__global__ __launch_bounds__(kBlock) void se_fold_kernel(const double* __restrict__ partials, int nblk,
                                                         double* __restrict__ out) {
    __shared__ double smem[kBlock / kWave];
    const size_t pt = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) s += partials[pt * nblk + i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[pt] = s;
}

// ---------------------------------------------------------------- host wrappers
// Device scratch layout inside ctx->ws: [params 4 x count][out count][nodes 2n][partials]
static int se_stage(tramp_ctx* ctx, size_t count, const double* const* host_in, int nin, double** dev_in,
                    double** dev_out, size_t extra_doubles, double** extra) {
    const size_t need = ((size_t)(nin + 1) * count + extra_doubles) * sizeof(double);
    int rc = tramp_ws_reserve(ctx, need);
    if (rc) return rc;
    double* p = ctx->ws;
    for (int i = 0; i < nin; ++i) {
        dev_in[i] = p;
        TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(p, host_in[i], count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        p += count;
    }
    *dev_out = p;
    p += count;
    if (extra) *extra = p;
    return TRAMP_OK;
}

static int se_finish(tramp_ctx* ctx, size_t count, const double* dev_out, double* host_out) {
    TRAMP_LAUNCH_CHECK(ctx);
    TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(host_out, dev_out, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TRAMP_OK;
}

extern "C" int tramp_se_gb_prior(tramp_ctx* ctx, size_t count, const double* ax, const double* rho,
                                 const double* mean, const double* var, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax && rho && mean && var && v);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i)
        TRAMP_REQUIRE(ctx, ax[i] >= 0.0 && rho[i] > 0.0 && rho[i] < 1.0 && var[i] > 0.0);
    const double* hin[4] = {ax, rho, mean, var};
    double *din[4], *dout;
    int rc = se_stage(ctx, count, hin, 4, din, &dout, 0, nullptr);
    if (rc) return rc;
    Se1dParams P{din[0], din[1], din[2], din[3], tmath::Pwl3Spec{}};
    hipLaunchKernelGGL(se_1d_kernel, dim3((unsigned)count), dim3(kBlock), 0, ctx->stream, 0, P, dout);
    g_se_evals = (unsigned long long)count * 2 * 8 * kPanels1D;
    return se_finish(ctx, count, dout, v);
}

// ---------------------------------------------------------------- K15: Marchenko-Pastur linear layer
// AnalyticalLinearChannel.compute_n_eff / compute_backward_error / compute_forward_error
// (tramp/channels/linear/analytical_linear_channel.py:28-53) with the closed-form eta transform of the
// Marchenko-Pastur law (tramp/ensembles/marchenko_pastur_ensemble.py:39-45).  One thread per point.
__global__ void se_mp_kernel(size_t count, int dir, const double* __restrict__ alpha_d,
                             const double* __restrict__ mean_spectrum_d, const double* __restrict__ az_d,
                             const double* __restrict__ ax_d, double* __restrict__ v_out, double* __restrict__ n_eff_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double alpha = alpha_d[i], ax = ax_d[i];
    double az = az_d[i];
    if (dir == TRAMP_BWD) az = fmax(1e-11, az);                          // :43
    double n_eff;
    if (ax == 0.0) n_eff = 0.0;                                          // :30-32
    else if (az / ax == 0.0) n_eff = fmin(1.0, alpha);                   // :33-35
    else {
        const double gamma = ax / az;
        const double sa = sqrt(alpha);
        const double zmax = (1.0 + sa) * (1.0 + sa), zmin = (1.0 - sa) * (1.0 - sa);
        const double root = sqrt(gamma * zmax + 1.0) - sqrt(gamma * zmin + 1.0);
        const double F = root * root;                                    // marchenko_pastur_ensemble.py:39-41
        n_eff = 1.0 - (1.0 - F / (4.0 * gamma));                         // :36-37 with eta = 1 - F / (4 gamma)
    }
    double v;
    if (dir == TRAMP_BWD) v = (1.0 - n_eff) / az;                        // :44-46
    else v = (ax == 0.0) ? mean_spectrum_d[i] / (alpha * az) : n_eff / (alpha * ax);   // :48-53
    v_out[i] = v;
    if (n_eff_out) n_eff_out[i] = n_eff;
}

extern "C" int tramp_se_mp_channel(tramp_ctx* ctx, int dir, size_t count, const double* alpha,
                                   const double* mean_spectrum, const double* az, const double* ax, double* v,
                                   double* n_eff) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, alpha && mean_spectrum && az && ax && v);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, alpha[i] > 0.0);
    const double* hin[4] = {alpha, mean_spectrum, az, ax};
    double *din[4], *dout, *dout2 = nullptr;
    int rc = se_stage(ctx, count, hin, 4, din, &dout, count, &dout2);
    if (rc) return rc;
    hipLaunchKernelGGL(se_mp_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, count, dir, din[0],
                       din[1], din[2], din[3], dout, dout2);
    if (n_eff)
        TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(n_eff, dout2, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return se_finish(ctx, count, dout, v);
}

extern "C" int tramp_se_binary_prior(tramp_ctx* ctx, size_t count, const double* ax, const double* p_pos, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax && p_pos && v);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, ax[i] >= 0.0 && p_pos[i] > 0.0 && p_pos[i] < 1.0);
    const double* hin[2] = {ax, p_pos};
    double *din[2], *dout;
    int rc = se_stage(ctx, count, hin, 2, din, &dout, 0, nullptr);
    if (rc) return rc;
    Se1dParams P{din[0], din[1], nullptr, nullptr, tmath::Pwl3Spec{}};
    hipLaunchKernelGGL(se_1d_kernel, dim3((unsigned)count), dim3(kBlock), 0, ctx->stream, 4, P, dout);
    g_se_evals = (unsigned long long)count * 2 * 8 * kPanels1D;
    return se_finish(ctx, count, dout, v);
}

extern "C" int tramp_se_gm_prior(tramp_ctx* ctx, size_t count, const double* ax, int K, const double* probs,
                                 const double* means, const double* vars, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, ax && probs && means && vars && v && K >= 1 && K <= tmath::kGmMax);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, ax[i] >= 0.0);
    const double* hin[1] = {ax};
    double *din[1], *dout;
    int rc = se_stage(ctx, count, hin, 1, din, &dout, 0, nullptr);
    if (rc) return rc;
    Se1dParams P{din[0], nullptr, nullptr, nullptr, tmath::Pwl3Spec{}};
    P.gm.K = K;
    for (int k = 0; k < K; ++k) {
        TRAMP_REQUIRE(ctx, probs[k] > 0.0 && vars[k] > 0.0);
        P.gm.probs[k] = probs[k];
        P.gm.means[k] = means[k];
        P.gm.vars[k] = vars[k];
    }
    hipLaunchKernelGGL(se_1d_kernel, dim3((unsigned)count), dim3(kBlock), 0, ctx->stream, 5, P, dout);
    g_se_evals = (unsigned long long)count * K * 8 * kPanels1D;
    return se_finish(ctx, count, dout, v);
}

extern "C" int tramp_se_sgn_lik(tramp_ctx* ctx, size_t count, const double* az, const double* tau_z, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az && tau_z && v);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, az[i] - 1.0 / tau_z[i] > 0.0);   // sgn_likelihood.py:80-81
    const double* hin[2] = {az, tau_z};
    double *din[2], *dout;
    int rc = se_stage(ctx, count, hin, 2, din, &dout, 0, nullptr);
    if (rc) return rc;
    Se1dParams P{din[0], din[1], nullptr, nullptr, tmath::Pwl3Spec{}};
    hipLaunchKernelGGL(se_1d_kernel, dim3((unsigned)count), dim3(kBlock), 0, ctx->stream, 1, P, dout);
    g_se_evals = (unsigned long long)count * 1 * 8 * kPanels1D;
    return se_finish(ctx, count, dout, v);
}

extern "C" int tramp_se_relu_lik(tramp_ctx* ctx, size_t count, const double* az, const double* tau_z, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az && tau_z && v);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, az[i] - 1.0 / tau_z[i] > 0.0);   // piecewise_linear_likelihood.py:123-124
    const double* hin[2] = {az, tau_z};
    double *din[2], *dout;
    int rc = se_stage(ctx, count, hin, 2, din, &dout, 0, nullptr);
    if (rc) return rc;
    Se1dParams P{din[0], din[1], nullptr, nullptr, tmath::Pwl3Spec{}};
    hipLaunchKernelGGL(se_1d_kernel, dim3((unsigned)count), dim3(kBlock), 0, ctx->stream, 2, P, dout);
    g_se_evals = (unsigned long long)count * 1 * 8 * kPanels1D;
    return se_finish(ctx, count, dout, v);
}

extern "C" int tramp_se_pwl3_lik(tramp_ctx* ctx, size_t count, double z_lo, double z_hi, const double* slopes3,
                                 const double* x03, const double* az, const double* tau_z, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az && tau_z && v && slopes3 && x03 && z_lo < z_hi);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, az[i] - 1.0 / tau_z[i] > 0.0);   // :123-124
    const double* hin[2] = {az, tau_z};
    double *din[2], *dout;
    int rc = se_stage(ctx, count, hin, 2, din, &dout, 0, nullptr);
    if (rc) return rc;
    Se1dParams P{din[0], din[1], nullptr, nullptr,
                 tmath::Pwl3Spec{z_lo, z_hi, {slopes3[0], slopes3[1], slopes3[2]}, {x03[0], x03[1], x03[2]}}};
    hipLaunchKernelGGL(se_1d_kernel, dim3((unsigned)count), dim3(kBlock), 0, ctx->stream, 3, P, dout);
    g_se_evals = (unsigned long long)count * 2 * 8 * kPanels1D;
    return se_finish(ctx, count, dout, v);
}

static int panels2d() {
    const char* e = getenv("TRAMP_SE_PANELS2D");
    if (e) {
        int p = atoi(e);
        if (p >= 8 && p <= 4096) return p;
    }
    return g_panels2d;
}

// ---------------------------------------------------------------- K14 adaptive: panel refinement
// The brute-force tensor grid above spends 3.8e7 integrand evaluations per value whatever the
// integrand looks like; the reference's adaptive dblquad needs ~4e4 (tramp/utils/integration.py:31-73).
// Adaptive cubature in GPU form: a work list of square panels, one workgroup per panel.  The 256
// threads evaluate the 8 x 8 Gauss-Legendre nodes of the panel's FOUR children (one wavefront per
// child: its integral is one 64-lane shuffle reduction); the children's sum is compared with the
// estimate the panel itself got one level up.  Agreement to the tolerance: the children's sum is
// accepted (composite GL-8 gains 2^16 per halving, so it is far better than the difference says).
// Otherwise the four children go on the next list with their own estimates.  Lists are compacted by
// a prefix scan in one workgroup, accepted values are folded in list order: the result does not depend
// on scheduling (run-to-run bit-identical).  The transition lines of the piecewise-linear posterior
// (r0 = 0, bz = 0: linear_region.py:83-101) end up covered by small panels, everything else by a few
// large ones: typically 1e5 - 4e5 evaluations instead of 3.8e7.
struct SePanel {
    double x0, y0, h, est;      // lower-left corner, edge, integral estimate from the parent level
};

constexpr int kSeListCap = 1 << 16;
constexpr int kSeMaxDepth = 14;

template <int KIND, bool FWD>
__global__ __launch_bounds__(kBlock) void se_2d_refine_kernel(Se2dParams P, int pt, const SePanel* __restrict__ in,
                                                              int n_in, double tol, int last_level,
                                                              SePanel* __restrict__ children, int* __restrict__ flags,
                                                              double* __restrict__ accepted) {
    __shared__ double child_int[4];
    const int e = blockIdx.x;
    if (e >= n_in) return;
    const SePanel p = in[e];
    const double az = P.az[pt], ax = (KIND == 3 || KIND == 6) ? 0.0 : P.ax[pt], tau = P.tau[pt];
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double hc = 0.5 * p.h;                                  // child edge
    const double cx0 = p.x0 + (c & 1) * hc, cy0 = p.y0 + (c >> 1) * hc;
    const int gi = lane >> 3, gj = lane & 7;
    const double x1 = cx0 + 0.5 * hc * (1.0 + kGLx[gi]);
    const double x2 = cy0 + 0.5 * hc * (1.0 + kGLx[gj]);
    const double w = 0.25 * hc * hc * kGLw[gi] * kGLw[gj] * norm_pdf(x1) * norm_pdf(x2);
    double f;
    if (KIND == 3) f = abs_lik_integrand(x1, x2, az, tau);
    else if (KIND == 6) f = modulus_lik_integrand(x1, x2, az, tau);
    else if (KIND == 5) f = pwl3_integrand<FWD>(x1, x2, az, ax, tau, P.spec3);
    else f = pwl_integrand<(KIND == 3 || KIND == 5 || KIND == 6 ? 0 : KIND), FWD>(x1, x2, az, ax, tau, P.spec);
    double v = w * f;
    v = wave_sum(v);
    if (lane == 0) child_int[c] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s4 = (child_int[0] + child_int[1]) + (child_int[2] + child_int[3]);
        const double err = fabs(s4 - p.est);
        const bool ok = (err <= tol) || last_level || !(s4 == s4);       // NaN: accept, it poisons the result as it must
        flags[e] = ok ? 0 : 1;
        accepted[e] = ok ? s4 : 0.0;
        if (!ok) {
            for (int k = 0; k < 4; ++k)
                children[4 * (size_t)e + k] = SePanel{p.x0 + (k & 1) * hc, p.y0 + (k >> 1) * hc, hc, child_int[k]};
        }
    }
}

// one workgroup: next list = children of the flagged panels, in list order; *n_out = its length;
// total += sum of accepted[] in list order
__global__ __launch_bounds__(kBlock) void se_2d_compact_kernel(const SePanel* __restrict__ children,
                                                               const int* __restrict__ flags,
                                                               const double* __restrict__ accepted, int n_in,
                                                               SePanel* __restrict__ out, int cap, int* __restrict__ n_out,
                                                               double* __restrict__ total) {
    __shared__ int s_cnt[kBlock];
    __shared__ double s_sum[kBlock / kWave];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    double acc = 0.0;
    __syncthreads();
    for (int start = 0; start < n_in; start += kBlock) {
        const int i = start + threadIdx.x;
        const int fl = (i < n_in) ? flags[i] : 0;
        if (i < n_in) acc += accepted[i];
        s_cnt[threadIdx.x] = fl;
        __syncthreads();
        // inclusive scan (Hillis-Steele) over the 256 flags
        for (int off = 1; off < kBlock; off <<= 1) {
            int v = (threadIdx.x >= off) ? s_cnt[threadIdx.x - off] : 0;
            __syncthreads();
            s_cnt[threadIdx.x] += v;
            __syncthreads();
        }
        const int pos = s_base + s_cnt[threadIdx.x] - fl;
        if (fl && 4 * (pos + 1) <= cap)
            for (int k = 0; k < 4; ++k) out[4 * pos + k] = children[4 * (size_t)i + k];
        __syncthreads();
        if (threadIdx.x == kBlock - 1) s_base += s_cnt[kBlock - 1];
        __syncthreads();
    }
    // fixed-order fold: thread-strided partial sums, then the usual tree
    acc = block_sum(acc, s_sum);
    if (threadIdx.x == 0) {
        *n_out = 4 * s_base;
        *total += acc;
    }
}

__global__ void se_2d_seed_kernel(int p0, double lo, double h, SePanel* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p0 * p0) return;
    // est = +inf: the first pass refines every seed panel unconditionally
    out[i] = SePanel{lo + (i % p0) * h, lo + (i / p0) * h, h, __longlong_as_double(0x7ff0000000000000LL)};
}


extern "C" int tramp_se_last_evals(tramp_ctx* ctx, unsigned long long* evals) {
    if (!ctx || !evals) return TRAMP_EINVAL;
    *evals = g_se_evals;
    return TRAMP_OK;
}

template <int KIND, bool FWD>
static int se_2d_brute(tramp_ctx* ctx, size_t count, const double* az, const double* ax, const double* tau,
                       double* v, const tmath::PwlSpec& spec, const tmath::Pwl3Spec& spec3) {
    const int panels = panels2d();
    const int n = panels * 8;
    const int strip = 8;
    const int nstrips = (n + strip - 1) / strip;
    const double* hin[3] = {az, ax ? ax : az, tau};
    double *din[3], *dout, *extra;
    int rc = se_stage(ctx, count, hin, 3, din, &dout, 2 * (size_t)n + count * (size_t)nstrips, &extra);
    if (rc) return rc;
    double* nodes = extra;
    double* partials = extra + 2 * (size_t)n;
    hipLaunchKernelGGL(se_nodes_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, panels, nodes);
    Se2dParams P{din[0], din[1], din[2], spec, spec3};
    hipLaunchKernelGGL((se_2d_kernel<KIND, FWD>), dim3(nstrips, (unsigned)count), dim3(kBlock), 0, ctx->stream, P, nodes,
                       n, strip, partials);
    hipLaunchKernelGGL(se_fold_kernel, dim3((unsigned)count), dim3(kBlock), 0, ctx->stream, partials, nstrips, dout);
    g_se_evals += (unsigned long long)count * n * n;
    return se_finish(ctx, count, dout, v);
}

template <int KIND, bool FWD>
static int se_2d_run(tramp_ctx* ctx, size_t count, const double* az, const double* ax, const double* tau,
                     double* v, const tmath::PwlSpec& spec = tmath::PwlSpec{},
                     const tmath::Pwl3Spec& spec3 = tmath::Pwl3Spec{}) {
    g_se_evals = 0;
    static const int mode_env = getenv("TRAMP_SE_2D") ? atoi(getenv("TRAMP_SE_2D")) : 1;     // 0: brute-force grid
    if (!mode_env) return se_2d_brute<KIND, FWD>(ctx, count, az, ax, tau, v, spec, spec3);
    // device scratch: params, two panel lists, children, flags, accepted, counters
    const size_t cap = kSeListCap;
    const double* hin[3] = {az, ax ? ax : az, tau};
    double *din[3], *dout, *extra;
    const size_t panel_d = sizeof(SePanel) / sizeof(double);
    const size_t extra_d = (2 * cap + 4 * cap) * panel_d + cap /*flags (as doubles: over-allocated)*/ + cap + 8;
    int rc = se_stage(ctx, count, hin, 3, din, &dout, extra_d, &extra);
    if (rc) return rc;
    SePanel* listA = reinterpret_cast<SePanel*>(extra);
    SePanel* listB = listA + cap;
    SePanel* children = listB + cap;
    int* flags = reinterpret_cast<int*>(children + 4 * cap);
    double* accepted = reinterpret_cast<double*>(flags) + cap;
    double* total = accepted + cap;
    int* n_out = reinterpret_cast<int*>(total + 1);
    Se2dParams P{din[0], din[1], din[2], spec, spec3};
    const int p0 = 8;                              // seed: 8 x 8 panels of edge 2.5 on [-10, 10]^2
    std::vector<size_t> redo;
    for (size_t pt = 0; pt < count; ++pt) {
        TRAMP_HIP_CHECK(ctx, hipMemsetAsync(total, 0, sizeof(double), ctx->stream));
        hipLaunchKernelGGL(se_2d_seed_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, p0, -10.0, 20.0 / p0, listA);
        int n_in = p0 * p0;
        SePanel *cur = listA, *nxt = listB;
        double tol = 0.0;                          // pass 0 refines everything; its sum sets the scale
        bool overflow = false;
        for (int depth = 0; depth <= kSeMaxDepth && n_in > 0; ++depth) {
            const int last = depth == kSeMaxDepth;
            hipLaunchKernelGGL((se_2d_refine_kernel<KIND, FWD>), dim3(n_in), dim3(kBlock), 0, ctx->stream, P, (int)pt, cur,
                               n_in, tol, last, children, flags, accepted);
            hipLaunchKernelGGL(se_2d_compact_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, children, flags, accepted,
                               n_in, nxt, (int)cap, n_out, total);
            g_se_evals += (unsigned long long)n_in * 256;
            int n_next = 0;
            double first_sum = 0.0;
            TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(&n_next, n_out, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            if (depth == 0) {
                // scale of the integral: the children estimates of the seed panels
                static thread_local SePanel host_children[4 * 64];
                TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(host_children, nxt, sizeof(SePanel) * 4 * p0 * p0, hipMemcpyDeviceToHost,
                                                    ctx->stream));
                TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                for (int i = 0; i < 4 * p0 * p0; ++i) first_sum += host_children[i].est;
                tol = 1e-10 * fabs(first_sum) + 1e-300;
                if (!(first_sum == first_sum)) tol = 0.0;
            } else {
                TRAMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            }
            if ((size_t)n_next > cap) {
                overflow = true;
                break;
            }
            n_in = n_next;
            SePanel* t = cur;
            cur = nxt;
            nxt = t;
        }
        if (overflow) {
            redo.push_back(pt);
            continue;
        }
        TRAMP_HIP_CHECK(ctx, hipMemcpyAsync(dout + pt, total, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    rc = se_finish(ctx, count, dout, v);
    if (rc) return rc;
    // a point whose list outgrew the buffers (pathologically sharp integrand) falls back to the dense grid
    for (size_t pt : redo) {
        double vv = 0.0;
        rc = se_2d_brute<KIND, FWD>(ctx, 1, az + pt, ax ? ax + pt : nullptr, tau + pt, &vv, spec, spec3);
        if (rc) return rc;
        v[pt] = vv;
    }
    return TRAMP_OK;
}

extern "C" int tramp_se_abs_lik(tramp_ctx* ctx, size_t count, const double* az, const double* tau_z, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az && tau_z && v);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, az[i] - 1.0 / tau_z[i] > 0.0);   // abs_likelihood.py:61-62
    return se_2d_run<3, false>(ctx, count, az, nullptr, tau_z, v);
}

extern "C" int tramp_se_modulus_lik(tramp_ctx* ctx, size_t count, const double* az, const double* tau_z, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az && tau_z && v);
    if (!count) return TRAMP_OK;
    // u_eff = max(0, az tau - 1) = 0 (modulus_likelihood.py:103-110): the 1-D average of sqrt(2 pi az) relu(y) y^2 / 2 over
    // y = xi / sqrt(az) is 1 / az in closed form (int_0^inf xi^3 N(xi) = 2 / sqrt(2 pi)); those points never reach the device
    std::vector<double> a2, t2;
    std::vector<size_t> idx;
    for (size_t i = 0; i < count; ++i) {
        TRAMP_REQUIRE(ctx, az[i] > 0.0 && tau_z[i] > 0.0);
        if (az[i] * tau_z[i] - 1.0 > 0.0) {
            a2.push_back(az[i]);
            t2.push_back(tau_z[i]);
            idx.push_back(i);
        } else {
            v[i] = 1.0 / az[i];
        }
    }
    if (idx.empty()) return TRAMP_OK;
    std::vector<double> out(idx.size());
    int rc = se_2d_run<6, false>(ctx, idx.size(), a2.data(), nullptr, t2.data(), out.data());
    if (rc) return rc;
    for (size_t j = 0; j < idx.size(); ++j) v[idx[j]] = out[j];
    return TRAMP_OK;
}

extern "C" int tramp_se_pwl_channel(tramp_ctx* ctx, int kind, int dir, size_t count, const double* az,
                                    const double* ax, const double* tau_z, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az && ax && tau_z && v);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, az[i] > 0.0 && ax[i] >= 0.0 && tau_z[i] > 0.0);
    const bool f = dir == TRAMP_FWD;
    switch (kind) {
        case TRAMP_SGN: return f ? se_2d_run<0, true>(ctx, count, az, ax, tau_z, v) : se_2d_run<0, false>(ctx, count, az, ax, tau_z, v);
        case TRAMP_ABS: return f ? se_2d_run<1, true>(ctx, count, az, ax, tau_z, v) : se_2d_run<1, false>(ctx, count, az, ax, tau_z, v);
        case TRAMP_RELU: return f ? se_2d_run<2, true>(ctx, count, az, ax, tau_z, v) : se_2d_run<2, false>(ctx, count, az, ax, tau_z, v);
    }
    return tramp_fail(ctx, TRAMP_EINVAL, "invalid argument", "kind");
}

extern "C" int tramp_se_pwl2_channel(tramp_ctx* ctx, int dir, size_t count, double slope_pos, double slope_neg,
                                     double x0_pos, double x0_neg, double zcut, const double* az,
                                     const double* ax, const double* tau_z, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az && ax && tau_z && v);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, az[i] > 0.0 && ax[i] >= 0.0 && tau_z[i] > 0.0);
    tmath::PwlSpec spec{{slope_pos, slope_neg}, {x0_pos, x0_neg}, zcut};
    return dir == TRAMP_FWD ? se_2d_run<4, true>(ctx, count, az, ax, tau_z, v, spec)
                            : se_2d_run<4, false>(ctx, count, az, ax, tau_z, v, spec);
}

extern "C" int tramp_se_pwl3_channel(tramp_ctx* ctx, int dir, size_t count, double z_lo, double z_hi,
                                     const double* slopes3, const double* x03, const double* az,
                                     const double* ax, const double* tau_z, double* v) {
    if (!ctx) return TRAMP_EINVAL;
    TRAMP_REQUIRE(ctx, az && ax && tau_z && v && slopes3 && x03 && z_lo < z_hi);
    TRAMP_REQUIRE(ctx, dir == TRAMP_FWD || dir == TRAMP_BWD);
    if (!count) return TRAMP_OK;
    for (size_t i = 0; i < count; ++i) TRAMP_REQUIRE(ctx, az[i] > 0.0 && ax[i] >= 0.0 && tau_z[i] > 0.0);
    tmath::Pwl3Spec spec{z_lo, z_hi, {slopes3[0], slopes3[1], slopes3[2]}, {x03[0], x03[1], x03[2]}};
    return dir == TRAMP_FWD ? se_2d_run<5, true>(ctx, count, az, ax, tau_z, v, tmath::PwlSpec{}, spec)
                            : se_2d_run<5, false>(ctx, count, az, ax, tau_z, v, tmath::PwlSpec{}, spec);
}
