"""CPU oracle for the Tree-AMP message-passing hot path.

TEST INFRASTRUCTURE ONLY.  This module is a NumPy/SciPy restatement of the
reference algorithm (sphinxteam/tramp) for the hot path of SURVEY.md section 8.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``tramp_amd``) never imports it and has no CPU fallback.

Parity pinning: every function below is checked in ``tests/test_oracle_golden.py``
against golden vectors captured by importing the reference in the build
container (``tools/make_golden.py`` -> ``tests/golden/*.npz``), including the
known-answer values listed in SURVEY.md section 8(c).  The reference's own tests hold
no fixture files for this path; LinearChannel has no reference test at all, so
for it the oracle is additionally pinned on the algebraic identity
``rz = (az I + ax W^T W)^-1 (bz + W^T bx)``.

Third-party arithmetic not vendored in the reference:
``scipy.special.{erfcx, erf, expit, softmax}``, ``numpy.logaddexp``,
``numpy.linalg.svd``, ``scipy.integrate.{quad, dblquad}`` (numpy 2.2.6 /
scipy 1.15.3 in the build image).

All ``file:line`` citations are relative to the reference repository root.
"""
import numpy as np
from scipy.special import erfcx, erf, expit
from scipy.integrate import quad, dblquad

AMIN = 1e-11   # tramp/base.py:238-239
AMAX = 1e+11

SQRT2 = np.sqrt(2)


# --------------------------------------------------------------------------
# A1  node protocol arithmetic                      tramp/base.py:44-46,245-255
# --------------------------------------------------------------------------
def inv(v):
    "Safe inverse, floor 1e-20 (tramp/base.py:44-46)."
    return 1 / np.maximum(v, 1e-20)


def a_new(v, a):
    "tramp/base.py:245-248"
    return np.clip(inv(v) - a, AMIN, AMAX)


def ab_new(r, v, a, b):
    "tramp/base.py:250-255"
    an = np.clip(inv(v) - a, AMIN, AMAX)
    bn = r * (a + an) - b
    return an, bn


# --------------------------------------------------------------------------
# A2  Gauss-Bernoulli prior      tramp/priors/gauss_bernoulli_prior.py:26-83
#                                tramp/beliefs/sparse.py:5-27, normal.py:3-4
# --------------------------------------------------------------------------
def normal_A(a, b):
    "tramp/beliefs/normal.py:3-4"
    return 0.5 * (b**2 / a + np.log(2 * np.pi / a))


def gb_params(rho, mean, var):
    "Natural parameters and eta of the prior (gauss_bernoulli_prior.py:32-36)."
    a0 = 1 / var
    b0 = mean / var
    eta = normal_A(a0, b0) - np.log(rho / (1 - rho))
    return a0, b0, eta


def gb_prior_posterior(ax, bx, rho=0.5, mean=0, var=1, isotropic=True):
    """(rx, vx) of the Gauss-Bernoulli denoiser.

    gauss_bernoulli_prior.py:70-77 with sparse.p/r/v (beliefs/sparse.py:9-22).
    """
    a0, b0, eta = gb_params(rho, mean, var)
    a = ax + a0
    b = bx + b0
    s = expit(normal_A(a, b) - eta)
    rx = s * (b / a)
    vx = s / a + s * (1 - s) * (b / a)**2
    if isotropic:
        vx = vx.mean()
    return rx, vx


def gb_prior_message(ax, bx, rho=0.5, mean=0, var=1, isotropic=True):
    "Prior.compute_forward_message (tramp/priors/base_prior.py:13-16)."
    rx, vx = gb_prior_posterior(ax, bx, rho, mean, var, isotropic)
    return ab_new(rx, vx, ax, bx)


def gb_prior_log_partition(ax, bx, rho=0.5, mean=0, var=1):
    "gauss_bernoulli_prior.py:79-83 (per-element, before the mean)."
    a0, b0, eta = gb_params(rho, mean, var)
    a = ax + a0
    b = bx + b0
    return (np.logaddexp(eta, normal_A(a, b))
            - np.logaddexp(eta, normal_A(a0, b0)))


# --------------------------------------------------------------------------
# A3  MAP L1 prior (soft threshold)   tramp/priors/map_L1_norm_prior.py:11-71
# --------------------------------------------------------------------------
def l1_support_mask(bx, gamma):
    "v_soft_threshold (map_L1_norm_prior.py:16-17): boolean, bit-exact row."
    return np.abs(bx) > gamma


def l1_prior_posterior(ax, bx, gamma=1, isotropic=True):
    "map_L1_norm_prior.py:66-71"
    with np.errstate(divide="ignore", invalid="ignore"):
        shrink = np.maximum(0, 1 - gamma / np.abs(bx)) * bx
    rx = (1 / ax) * shrink
    vx = (1 / ax) * l1_support_mask(bx, gamma)
    if isotropic:
        vx = vx.mean()
    return rx, vx


def l1_prior_message(ax, bx, gamma=1, isotropic=True):
    rx, vx = l1_prior_posterior(ax, bx, gamma, isotropic)
    return ab_new(rx, vx, ax, bx)


# --------------------------------------------------------------------------
# A4 / A5  Gaussian prior and Gaussian likelihood: constant messages
#   tramp/priors/gaussian_prior.py:63-68,76-79,86-93
#   tramp/likelihoods/gaussian_likelihood.py:43-49,58-61,68-75
# --------------------------------------------------------------------------
# --------------------------------------------------------------------------
# Widening: group-sparsity MAP prior (the total-variation prior on a gradient field)
#   tramp/priors/map_L21_norm_prior.py:7-20,69-74 (axis = 0)
# --------------------------------------------------------------------------
def l21_support_mask(bx, gamma):
    return np.linalg.norm(bx, axis=0) > gamma


def l21_prior_posterior(ax, bx, gamma=1):
    with np.errstate(all="ignore"):
        norm_k = np.linalg.norm(bx, axis=0, keepdims=True)
        rx = (1 / ax) * (np.maximum(0, 1 - gamma / norm_k) * bx)                       # :12-14
        norm = np.linalg.norm(bx, axis=0, keepdims=False)
        v = (norm > gamma) * (1 + (bx**2 / norm**2 - 1) * gamma / norm)                # :17-20
    return rx, ((1 / ax) * v).mean()


def l21_prior_message(ax, bx, gamma=1):
    rx, vx = l21_prior_posterior(ax, bx, gamma)
    return ab_new(rx, vx, ax, bx)


def gauss_prior_posterior(ax, bx, mean=0, var=1):
    a = ax + 1 / var
    b = bx + mean / var
    return b / a, 1 / a


def gauss_prior_message(ax, bx, mean=0, var=1):
    "Bypasses ab_new entirely (gaussian_prior.py:86-89)."
    return (1 / var) * np.ones_like(ax), (mean / var) * np.ones_like(bx)


def gauss_prior_error(ax, var=1):
    return 1 / (ax + 1 / var)


def gauss_lik_posterior(az, bz, y, var=1):
    a = az + 1 / var
    b = bz + (1 / var) * y
    return b / a, 1 / a


def gauss_lik_message(az, bz, y, var=1):
    "Constant (1/var, y/var) (gaussian_likelihood.py:68-71, :17)."
    return 1 / var, y / var


def gauss_lik_error(az, tau_z, var=1):
    return 1 / (az + 1 / var)


# --------------------------------------------------------------------------
# Half-line truncated normal    tramp/utils/truncated_normal.py:21-29,92-94,
#                               147-149,218-298
# side=+1 : z in [0, +inf)   side=-1 : z in (-inf, 0]
# --------------------------------------------------------------------------
def log_Phi(x):
    "log Phi(x) with the reference's x<30 mask (truncated_normal.py:21-29)."
    x = np.asarray(x, dtype=float)
    y = np.zeros_like(x)
    m = x < 30
    xm = x[m]
    y[m] = np.log(0.5 * erfcx(-xm / SQRT2)) - 0.5 * xm**2
    return y


def halfline_moments(r0, v0, side, zb=0):
    """Mean, variance and log-probability of N(r0, v0) restricted to a half line
    [zb, inf) (side=+1) or (-inf, zb] (side=-1); zb = 0 for Sgn/Abs/Relu.

    truncated_normal_mean/_var/_log_proba (truncated_normal.py:234-281) through
    the ``_inf`` branches G0_inf/G1_inf/G2_inf (:218-231), F1_inf/F2_inf
    (:92-94, :147-149).  For both sides the standardised bound is
    y0 = (0 - r0)/s0 and the reference evaluates the functions with s = side.
    """
    s0 = np.sqrt(v0)
    y0 = (zb - r0) / s0
    x = y0 / SQRT2
    e = erfcx(side * x)
    g1 = np.sqrt(2 / np.pi) * (side / e)
    g2 = (2 / np.sqrt(np.pi)) * (side * x / e)
    mean = r0 + s0 * g1
    var = v0 * (1. + g2 - g1**2)
    logp = log_Phi(-side * y0)
    return mean, var, logp


def halfline_logZ(r0, v0, side):
    "truncated_normal_logZ (truncated_normal.py:294-298)."
    _, _, g0 = halfline_moments(r0, v0, side)
    return 0.5 * np.log(2 * np.pi * v0) + 0.5 * r0**2 / v0 + g0


def norm_cdf(x):
    "tramp/utils/misc.py:50-52"
    return 0.5 * (1 + erf(x / SQRT2))


def norm_pdf(x):
    "tramp/utils/misc.py:55-57"
    return np.exp(-0.5 * x**2) / np.sqrt(2 * np.pi)


def halfline_proba(r0, v0, side, zb=0):
    "truncated_normal_proba (truncated_normal.py:284-291)."
    s0 = np.sqrt(v0)
    y0 = (zb - r0) / s0
    if side > 0:
        return norm_cdf(np.inf) - norm_cdf(y0)
    return norm_cdf(y0) - norm_cdf(-np.inf)


# --------------------------------------------------------------------------
# Finite-interval truncated normal (three-region channels)
#   tramp/utils/truncated_normal.py:13-16 (switch), :32-89 (F0), :97-144 (F1), :152-200 (F2),
#   :203-215 (G0/G1/G2), :234-298.  The branch masks inf/close/neg/pos/other are index masks.
# --------------------------------------------------------------------------
def _switch(x, y):
    "truncated_normal.py:13-18: order the pair so that |x| <= |y|"
    sw = np.abs(x) > np.abs(y)
    return np.where(sw, y, x), np.where(sw, x, y)


def pwl_log_Phi_masks(kind, az, bz, ax, bx):
    """The boolean index arrays `nonzero = (x < 30)` that log_Phi (tramp/utils/truncated_normal.py:21-29) takes for
    the (pos, neg) regions of a Sgn / Abs / Relu channel: x = -s (0 - r0)/sqrt(v0) with (r0, v0) from
    LinearRegion.get_r0_v0 (utils/linear_region.py:32-37) and s = +1 / -1 (G0_inf, :218-221), every operation as the
    reference orders it."""
    bz = np.asarray(bz, dtype=float)
    bx = np.asarray(bx, dtype=float)
    out = []
    for side, slope, x0 in PWL_REGIONS[kind]:
        a = az + slope**2 * ax
        b = bz + slope * (bx - ax * x0)
        r0, v0 = b / a, 1 / a
        y0 = (0 - r0) / np.sqrt(v0)
        out.append((-side * y0) < 30)
    return out


def interval_branch_masks(x, y, thresh=1e-7):
    """(close, neg, pos, other) of truncated_normal.py:68-82 for FINITE bounds, on the switched pair."""
    x, y = _switch(np.asarray(x, dtype=float), np.asarray(y, dtype=float))
    close = np.abs(x - y) <= thresh
    neg = (x < 0) & (y < 0)
    pos = (x > 0) & (y > 0)
    other = ~(neg | pos)
    return close, ~close & neg, ~close & pos, ~close & other


def _F012(x, y, thresh=1e-7):
    "F0, F1, F2 of truncated_normal.py for finite x, y (all three share the branch structure)."
    x, y = _switch(np.asarray(x, dtype=float), np.asarray(y, dtype=float))
    close, neg, pos, other = interval_branch_masks(x, y, thresh)
    F0 = np.zeros_like(x); F1 = np.zeros_like(x); F2 = np.zeros_like(x)
    with np.errstate(all="ignore"):
        # close (Taylor, :38-46, :102-110, :157-166) - restated literally, quirks included
        xc, yc = x[close], y[close]
        e = yc - xc
        F0[close] = (-xc * e + (1 / 6) * (xc**2 - 2) * e**2 - (1 / 180) * (xc**4 + 2 * xc**2 - 8)
                     + np.log(2 * e / np.sqrt(np.pi))) - xc**2
        F1[close] = np.sqrt(np.pi) * (xc + (1 / 2) * e - (1 / 6) * e**2 - (1 / 12) * e**3
                                      + (1 / 90) * xc * (xc**2 + 1.) * e**4)
        F2[close] = np.sqrt(np.pi) * (xc**2 - 1 / 2 + xc * e - (1 / 3) * (xc**2 - 1) * e**2
                                      - (1 / 3) * xc * e**3 + (1 / 90) * (2 * xc**4 + 3 * xc**2 - 8) * e**4)
        # neg (:49-54, :113-115, :169-171)
        xn, yn = x[neg], y[neg]
        D = np.exp(xn**2 - yn**2)
        den = D * erfcx(-yn) - erfcx(-xn)
        F0[neg] = np.log(np.abs(den)) - xn**2
        F1[neg] = (1 - D) / den
        F2[neg] = (xn - D * yn) / den
        # pos (:57-62, :118-120, :174-176)
        xp, yp = x[pos], y[pos]
        D = np.exp(xp**2 - yp**2)
        den = erfcx(xp) - D * erfcx(yp)
        F0[pos] = np.log(np.abs(den)) - xp**2
        F1[pos] = (1 - D) / den
        F2[pos] = (xp - D * yp) / den
        # other (:65-66, :123-125, :179-181)
        xo, yo = x[other], y[other]
        D = np.exp(xo**2 - yo**2)
        den = erf(yo) - erf(xo)
        F0[other] = np.log(np.abs(den))
        F1[other] = np.exp(-xo**2) * (1 - D) / den
        F2[other] = np.exp(-xo**2) * (xo - D * yo) / den
    return F0, F1, F2


def interval_moments(r0, v0, zmin, zmax):
    "Mean, variance, log-probability of N(r0, v0) on a FINITE [zmin, zmax] (truncated_normal.py:234-281)."
    r0 = np.asarray(r0, dtype=float)
    s0 = np.sqrt(v0)
    ymin = (zmin - r0) / s0
    ymax = (zmax - r0) / s0
    F0, F1, F2 = _F012(ymin / SQRT2, ymax / SQRT2)
    g0 = np.log(0.5) + F0
    g1 = np.sqrt(2 / np.pi) * F1
    g2 = (2 / np.sqrt(np.pi)) * F2
    return r0 + s0 * g1, v0 * (1. + g2 - g1**2), g0


def interval_proba(r0, v0, zmin, zmax):
    s0 = np.sqrt(v0)
    return norm_cdf((zmax - r0) / s0) - norm_cdf((zmin - r0) / s0)


# Three-region channels: [pos, mid, neg] as the reference constructors
#   tramp/channels/activation/piecewise_linear_channel.py:109-132
def pwl3_regions(kind, param=None):
    "[(zmin, zmax, slope, x0)] for pos, mid, neg"
    inf = np.inf
    if kind == "h-tanh":
        return [(1., inf, 0., 1.), (-1., 1., 1., 0.), (-inf, -1., 0., -1.)]
    if kind == "h-sigm":
        L = 2.5
        return [(L, inf, 0., 1.), (-L, L, 1 / (2 * L), 0.5), (-inf, -L, 0., 0.)]
    if kind == "door":
        w = float(param)
        return [(w, inf, 0., 1.), (-w, w, 0., -1.), (-inf, -w, 0., 1.)]
    raise KeyError(kind)


def _region_moments(zmin, zmax, r0, v0):
    if np.isinf(zmax):
        return halfline_moments(r0, v0, +1, zmin)
    if np.isinf(zmin):
        return halfline_moments(r0, v0, -1, zmax)
    return interval_moments(r0, v0, zmin, zmax)


def pwl3_channel_posterior(kind, direction, az, bz, ax, bx, isotropic=True, param=None):
    "PiecewiseLinearChannel.compute_*_posterior with three regions (merge_estimates :27-37)."
    bz = np.asarray(bz, dtype=float)
    bx = np.asarray(bx, dtype=float)
    rs, vs, As = [], [], []
    for zmin, zmax, slope, x0 in pwl3_regions(kind, param):
        a = az + slope**2 * ax
        b = bz + slope * (bx - ax * x0)
        r0, v0 = b / a, 1 / a
        rz, vz, g0 = _region_moments(zmin, zmax, r0, v0)
        A = 0.5 * np.log(2 * np.pi * v0) + 0.5 * r0**2 / v0 + g0 - 0.5 * ax * x0**2 + bx * x0
        rs.append(slope * rz + x0 if direction == "fwd" else rz)
        vs.append(slope**2 * vz if direction == "fwd" else vz)
        As.append(A)
    A = np.stack(As)
    with np.errstate(invalid="ignore"):
        e = np.exp(A - A.max(axis=0, keepdims=True))
        ps = e / e.sum(axis=0, keepdims=True)
    r = ps[0] * rs[0] + ps[1] * rs[1] + ps[2] * rs[2]
    Dr = (ps[0] * ps[1] * (rs[0] - rs[1])**2 + ps[0] * ps[2] * (rs[0] - rs[2])**2
          + ps[1] * ps[2] * (rs[1] - rs[2])**2)
    v = ps[0] * vs[0] + ps[1] * vs[1] + ps[2] * vs[2] + Dr
    if isotropic:
        v = v.mean()
    return r, v


def pwl3_channel_message(kind, direction, az, bz, ax, bx, param=None):
    r, v = pwl3_channel_posterior(kind, direction, az, bz, ax, bx, param=param)
    return ab_new(r, v, ax, bx) if direction == "fwd" else ab_new(r, v, az, bz)


def pwl3_mid_masks(kind, az, bz, ax, bx, param=None):
    "Branch masks (close, neg, pos, other) of the finite mid region, as the reference forms them."
    zmin, zmax, slope, x0 = pwl3_regions(kind, param)[1]
    a = az + slope**2 * ax
    b = np.asarray(bz, dtype=float) + slope * (np.asarray(bx, dtype=float) - ax * x0)
    r0, v0 = b / a, 1 / a
    s0 = np.sqrt(v0)
    return interval_branch_masks((zmin - r0) / s0 / SQRT2, (zmax - r0) / s0 / SQRT2)


def se_pwl3_channel_error(kind, direction, az, ax, tau_z, method="gl", param=None):
    "beliefs_measure summed over the three regions (linear_region.py:83-101)."
    def variance(bz, bx):
        return pwl3_channel_posterior(kind, direction, az, bz, ax, bx, isotropic=False, param=param)[1]
    u_eff = np.maximum(0, az * tau_z - 1)
    total = 0.
    for zmin, zmax, slope, x0 in pwl3_regions(kind, param):
        mean_x = ax * x0

        def integrand(bz, bx, zmin=zmin, zmax=zmax, slope=slope, x0=x0):
            a = az + slope**2 * ax
            b = bz + slope * (bx - ax * x0)
            return interval_proba(b / a, 1 / a, zmin, zmax) * variance(bz, bx)
        sx2 = ax * (slope**2 * ax * tau_z + 1)
        if ax == 0 or u_eff == 0 or slope == 0:
            total += gaussian_measure_2d(0, np.sqrt(az * u_eff), mean_x, np.sqrt(sx2), integrand, method)
        else:
            cov = np.array([[az * u_eff, slope * ax * u_eff], [slope * ax * u_eff, sx2]])
            total += gaussian_measure_2d_full(np.array([0, mean_x]), cov, integrand, method)
    return total


def pwl3_second_moment(kind, tau_z, param=None):
    tot = 0.
    for zmin, zmax, slope, x0 in pwl3_regions(kind, param):
        rz, vz, _ = _region_moments(zmin, zmax, np.float64(0), np.float64(tau_z))
        p = interval_proba(np.float64(0), np.float64(tau_z), zmin, zmax)
        tot += p * ((slope * rz + x0)**2 + slope**2 * vz)
    return tot


# --------------------------------------------------------------------------
# A6  Sgn likelihood     tramp/likelihoods/sgn_likelihood.py:23-41
#                        tramp/beliefs/positive.py:8-26
# --------------------------------------------------------------------------
def sgn_lik_posterior(az, bz, y, isotropic=True):
    b = bz * y
    m, v, _ = halfline_moments(b / az, 1 / az, +1)
    rz = y * m
    vz = v
    if isotropic:
        vz = vz.mean()
    return rz, vz


def sgn_lik_message(az, bz, y, isotropic=True):
    "Likelihood.compute_backward_message (base_likelihood.py:25-28)."
    rz, vz = sgn_lik_posterior(az, bz, y, isotropic)
    return ab_new(rz, vz, az, bz)


# --------------------------------------------------------------------------
# A7  Abs likelihood     tramp/likelihoods/abs_likelihood.py:22-40
#                        tramp/beliefs/binary.py:4-17
# --------------------------------------------------------------------------
def abs_lik_posterior(az, bz, y, isotropic=True):
    t = np.tanh(bz * y)
    rz = y * t
    vz = (y**2) * (1 - t**2)
    if isotropic:
        vz = vz.mean()
    return rz, vz


def abs_lik_message(az, bz, y, isotropic=True):
    rz, vz = abs_lik_posterior(az, bz, y, isotropic)
    return ab_new(rz, vz, az, bz)


# --------------------------------------------------------------------------
# section 8f rank 2: (Leaky)Relu likelihood
#   tramp/likelihoods/piecewise_linear_likelihood.py:9-63 (LinearRegionLikelihood),
#   :201-215 (compute_backward_posterior), :279-291 (region tables)
# --------------------------------------------------------------------------
def relu_lik_posterior(az, bz, y, slope_neg=0., isotropic=True):
    """Restates the reference region by region: a sloped region pins z = (y - x0)/slope when
    that z is strictly inside it (log-partition -inf otherwise); the flat region of Relu holds
    a truncated normal when y == x0 = 0; softmax of the log-partitions mixes them."""
    bz = np.asarray(bz, dtype=float)
    y = np.asarray(y, dtype=float)
    rs, vs, As = [], [], []
    for side, slope in ((+1, 1.), (-1, float(slope_neg))):
        if slope == 0:
            contains = (y == 0.)
            m, v, logp = halfline_moments(bz / az, 1 / az, side)
            r_k, v_k = m, v
            A_k = 0.5 * np.log(2 * np.pi / az) + 0.5 * (bz / az)**2 * az + logp
        else:
            z = y / slope
            contains = (z > 0) if side > 0 else (z < 0)
            r_k, v_k = z, np.zeros_like(z)
            A_k = -0.5 * az * (z**2) + bz * z - np.log(slope)
        rs.append(np.where(contains, r_k, 0))
        vs.append(np.where(contains, v_k, 0))
        As.append(np.where(contains, A_k, -np.inf))
    A = np.stack(As)
    with np.errstate(invalid="ignore"):
        e = np.exp(A - A.max(axis=0, keepdims=True))
        ps = e / e.sum(axis=0, keepdims=True)
    r = ps[0] * rs[0] + ps[1] * rs[1]
    v = ps[0] * vs[0] + ps[1] * vs[1] + ps[0] * ps[1] * (rs[0] - rs[1])**2
    if isotropic:
        v = v.mean()
    return r, v


def relu_lik_message(az, bz, y, slope_neg=0.):
    r, v = relu_lik_posterior(az, bz, y, slope_neg)
    return ab_new(r, v, az, bz)


def se_relu_lik_error(az, tau_z, method="quadpack"):
    "piecewise_linear_likelihood.py:120-146 for Relu: the sloped region has zero variance"
    mz_hat = az - 1 / tau_z
    assert mz_hat > 0
    sz_eff = np.sqrt(mz_hat + (mz_hat**2) * tau_z)

    def g(bz):
        bz = np.asarray(bz, dtype=float)
        _, v, _ = halfline_moments(bz / az, 1 / az, -1)
        return halfline_proba(bz / az, 1 / az, -1) * v
    return gaussian_measure(0, sz_eff, g, method)


# Three-region likelihoods (what to_observed turns an observed HardTanh / HardSigmoid /
# SymmetricDoor output into): tramp/likelihoods/piecewise_linear_likelihood.py:9-63 (regions),
# :156-218 (merge), :293-320 (constructors; NB HardSigmoidLikelihood uses l = 3 where the
# channel uses 2.5, restated as found).
def pwl3_lik_regions(kind, param=None):
    "[(zmin, zmax, slope, x0)] for pos, mid, neg"
    inf = np.inf
    if kind == "h-tanh":
        return [(1., inf, 0., 1.), (-1., 1., 1., 0.), (-inf, -1., 0., -1.)]
    if kind == "h-sigm":
        L = 3.
        return [(L, inf, 0., 1.), (-L, L, 1 / (2 * L), 0.5), (-inf, -L, 0., 0.)]
    if kind == "door":
        w = float(param)
        return [(w, inf, 0., 1.), (-w, w, 0., -1.), (-inf, -w, 0., 1.)]
    raise KeyError(kind)


def _region_proba(zmin, zmax, r0, v0):
    if np.isinf(zmax):
        return halfline_proba(r0, v0, +1, zmin)
    if np.isinf(zmin):
        return halfline_proba(r0, v0, -1, zmax)
    return interval_proba(r0, v0, zmin, zmax)


def pwl3_lik_posterior(kind, az, bz, y, isotropic=True, param=None):
    bz = np.asarray(bz, dtype=float)
    y = np.asarray(y, dtype=float) + np.zeros_like(bz)
    rs, vs, As = [], [], []
    for (zmin, zmax, slope, x0) in pwl3_lik_regions(kind, param):
        if slope == 0:
            contains = (y == x0)
            m, v, logp = _region_moments(zmin, zmax, bz / az, 1 / az)
            r_k, v_k = m, v
            A_k = 0.5 * np.log(2 * np.pi / az) + 0.5 * (bz / az)**2 * az + logp   # truncated_normal_logZ
        else:
            z = (y - x0) / slope
            contains = (zmin < z) & (z < zmax)
            r_k, v_k = z, np.zeros_like(z)
            A_k = -0.5 * az * (z**2) + bz * z - np.log(slope)
        rs.append(np.where(contains, r_k, 0))
        vs.append(np.where(contains, v_k, 0))
        As.append(np.where(contains, A_k, -np.inf))
    A = np.stack(As)
    with np.errstate(invalid="ignore"):
        e = np.exp(A - A.max(axis=0, keepdims=True))
        ps = e / e.sum(axis=0, keepdims=True)
    r = sum(p_ * r_ for p_, r_ in zip(ps, rs))
    v = sum(p_ * v_ for p_, v_ in zip(ps, vs))
    for i in range(3):
        for j in range(i + 1, 3):
            v = v + ps[i] * ps[j] * (rs[i] - rs[j])**2
    if isotropic:
        v = v.mean()
    return r, v


def pwl3_lik_message(kind, az, bz, y, param=None):
    r, v = pwl3_lik_posterior(kind, az, bz, y, param=param)
    return ab_new(r, v, az, bz)


def se_pwl3_lik_error(kind, az, tau_z, method="quadpack", param=None):
    """piecewise_linear_likelihood.py:120-146, 210-214: sum over regions of E[1_R f]; a sloped
    region pins z (variance 0), a flat one is a 1-D average over bz of p_R(bz) v(bz, y = x0)."""
    mz_hat = az - 1 / tau_z
    assert mz_hat > 0
    sz_eff = np.sqrt(mz_hat + (mz_hat**2) * tau_z)
    total = 0.
    for (zmin, zmax, slope, x0) in pwl3_lik_regions(kind, param):
        if slope != 0:
            continue

        def g(bz, zmin=zmin, zmax=zmax, x0=x0):
            bz = np.asarray(bz, dtype=float)
            _, v = pwl3_lik_posterior(kind, az, bz, x0, isotropic=False, param=param)
            return _region_proba(zmin, zmax, bz / az, 1 / az) * v
        total += gaussian_measure(0, sz_eff, g, method)
    return total


# --------------------------------------------------------------------------
# A8  Piecewise-linear channels (Sgn / Abs / Relu)
#   tramp/channels/activation/piecewise_linear_channel.py:27-53,80-106
#   tramp/utils/linear_region.py:32-64
# Region order is [pos, neg] as in the reference constructors.
# --------------------------------------------------------------------------
PWL_REGIONS = {
    # kind: [(side, slope, x0) for pos, (side, slope, x0) for neg]
    "sgn":  [(+1, 0., +1.), (-1, 0., -1.)],    # piecewise_linear_channel.py:80-84
    "abs":  [(+1, +1., 0.), (-1, -1., 0.)],    # :87-91
    "relu": [(+1, 1., 0.), (-1, 0., 0.)],      # :102-106
}


def pwl_regions(kind, param=None):
    """Region table [(side, slope, x0, zb) for pos, neg].  Besides the three hot-path kinds:
    'l-relu' (param = slope of the negative side; piecewise_linear_channel.py:72-77) and
    'a-abs' (param = shift of the kink; :94-99)."""
    if kind in PWL_REGIONS:
        return [r + (0.,) for r in PWL_REGIONS[kind]]
    if kind == "l-relu":
        return [(+1, 1., 0., 0.), (-1, float(param), 0., 0.)]
    if kind == "a-abs":
        return [(+1, +1., 0., float(param)), (-1, -1., 0., float(param))]
    raise KeyError(kind)


def pwl_region_terms(side, slope, x0, az, bz, ax, bx, zb=0):
    "Per-region (rz, vz, logZ): linear_region.py:32-64."
    a = az + slope**2 * ax
    b = bz + slope * (bx - ax * x0)
    r0 = b / a
    v0 = 1 / a
    rz, vz, g0 = halfline_moments(r0, v0, side, zb)
    logZ = (0.5 * np.log(2 * np.pi * v0) + 0.5 * r0**2 / v0 + g0
            - 0.5 * ax * x0**2 + bx * x0)
    return rz, vz, logZ


def pwl_channel_posterior(kind, direction, az, bz, ax, bx, isotropic=True, param=None):
    """(r, v) for direction 'fwd' (x side) or 'bwd' (z side).

    merge_estimates (piecewise_linear_channel.py:27-37) with scipy softmax over
    the two regions; the reference always takes v.mean().
    """
    bz = np.asarray(bz, dtype=float)
    bx = np.asarray(bx, dtype=float)
    rs, vs, As = [], [], []
    for side, slope, x0, zb in pwl_regions(kind, param):
        rz, vz, A = pwl_region_terms(side, slope, x0, az, bz, ax, bx, zb)
        if direction == "fwd":
            rs.append(slope * rz + x0)
            vs.append(slope**2 * vz)
        else:
            rs.append(rz)
            vs.append(vz)
        As.append(A)
    A = np.stack(As)
    e = np.exp(A - A.max(axis=0, keepdims=True))      # scipy.special.softmax
    ps = e / e.sum(axis=0, keepdims=True)
    r = ps[0] * rs[0] + ps[1] * rs[1]
    v = ps[0] * vs[0] + ps[1] * vs[1] + ps[0] * ps[1] * (rs[0] - rs[1])**2
    if isotropic:
        v = v.mean()
    return r, v


def pwl_channel_message(kind, direction, az, bz, ax, bx, param=None):
    "Channel.compute_forward/backward_message (channels/base_channel.py:9-17)."
    r, v = pwl_channel_posterior(kind, direction, az, bz, ax, bx, param=param)
    if direction == "fwd":
        return ab_new(r, v, ax, bx)
    return ab_new(r, v, az, bz)


def pwl_proba_ab(side, slope, x0, az, bz, ax, bx, zb=0):
    "LinearRegion.proba_ab (linear_region.py:78-81)."
    a = az + slope**2 * ax
    b = bz + slope * (bx - ax * x0)
    return halfline_proba(b / a, 1 / a, side, zb)


# --------------------------------------------------------------------------
# A9 / A10  Dense linear channel
#   tramp/channels/linear/linear_channel.py:8-15,30-46,55-125
# --------------------------------------------------------------------------
class LinearOracle:
    """x = W z with the reference's full-SVD formulation."""

    def __init__(self, W):
        W = np.asarray(W, dtype=float)
        self.W = W
        self.Nx, self.Nz = W.shape
        self.rank = np.linalg.matrix_rank(W)                     # :37
        self.alpha = self.Nx / self.Nz                           # :38
        U, s, VT = np.linalg.svd(W, full_matrices=True)          # :8-15
        S = np.zeros((U.shape[0], VT.shape[0]))
        S[:len(s), :len(s)] = np.diag(s)
        self.U, self.S, self.V = U, S, VT.T
        spectrum = np.zeros(self.Nz)                             # diag(S^T S), :41
        spectrum[:len(s)] = s**2
        self.spectrum = spectrum
        self.singular = spectrum[:self.rank]                     # :46

    def second_moment(self, tau_z):
        return tau_z * self.spectrum.sum() / self.Nx             # :55-56

    def n_eff(self, az, ax):
        ":58-67"
        if ax == 0:
            return 0.
        if az / ax == 0:
            return self.rank / self.Nz
        return np.sum(self.singular / (az / ax + self.singular)) / self.Nz

    def backward_mean(self, az, bz, ax, bx):
        ":69-78 (precompute_svd=True branch)"
        bx_svd = self.U.T @ bx
        bz_svd = self.V.T @ bz
        resolvent = 1 / (az + ax * self.spectrum)
        if bz.ndim > 1:
            resolvent = resolvent[:, np.newaxis]
        return self.V @ (resolvent * (bz_svd + self.S.T @ bx_svd))

    def backward_mean_solve(self, az, bz, ax, bx):
        ":79-82 (precompute_svd=False branch) - the algebraic identity."
        A = az * np.identity(self.Nz) + ax * (self.W.T @ self.W)
        return np.linalg.solve(A, bz + self.W.T @ bx)

    def forward_mean(self, az, bz, ax, bx):
        ":85-89"
        return self.W @ self.backward_mean(az, bz, ax, bx)

    def backward_variance(self, az, ax):
        ":91-97"
        az = np.maximum(1e-11, az)
        return (1 - self.n_eff(az, ax)) / az

    def forward_variance(self, az, ax):
        ":99-105"
        if ax == 0:
            return np.mean(self.singular) * self.rank / (self.Nx * az)
        return self.n_eff(az, ax) / (self.alpha * ax)

    def forward_message(self, az, bz, ax, bx):
        rx = self.forward_mean(az, bz, ax, bx)
        vx = self.forward_variance(az, ax)
        return ab_new(rx, vx, ax, bx)

    def backward_message(self, az, bz, ax, bx):
        rz = self.backward_mean(az, bz, ax, bx)
        vz = self.backward_variance(az, ax)
        return ab_new(rz, vz, az, bz)


class LinearSolveOracle(LinearOracle):
    """LinearChannel(W, precompute_svd=False) as the reference runs it (linear_channel.py:43-46, 79-82): C = W^T W,
    spectrum = eigvalsh(C) in ASCENDING order, singular = spectrum[:rank] (its smallest entries - the variances
    inherit that), means from np.linalg.solve, log-partition over the whole spectrum (:127-132)."""

    def __init__(self, W):
        W = np.asarray(W, dtype=float)
        self.W = W
        self.Nx, self.Nz = W.shape
        self.rank = np.linalg.matrix_rank(W)                     # :37
        self.alpha = self.Nx / self.Nz
        self.C = W.T @ W                                         # :43
        self.spectrum = np.linalg.eigvalsh(self.C)               # :44
        self.singular = self.spectrum[:self.rank]                # :46

    def backward_mean(self, az, bz, ax, bx):
        return self.backward_mean_solve(az, bz, ax, bx)          # :79-82

    def log_partition(self, az, bz, ax, bx):
        rz = self.backward_mean(az, bz, ax, bx)
        b = bz + self.W.T @ bx
        a = az + ax * self.spectrum
        return 0.5 * np.sum(b * rz) + 0.5 * np.sum(np.log(2 * np.pi / a))


class LinearThinOracle(LinearOracle):
    """Same results from the thin factorisation the HIP path stores.

    Only ``V_M`` (first ``M = Nx`` right singular vectors), ``U`` and the
    singular values enter.  With t = V_M^T bz and u = U^T bx,
    D = az + ax sigma^2:

      rz = bz/az + V_M [ (1/D - 1/az) t + sigma u / D ]      (Nx <  Nz)
      rz =         V   [ (t + sigma u) / D ]                  (Nx >= Nz)
      rx = U [ sigma (t + sigma u) / D ]

    This is what tramp_amd computes; the class exists so that tests can check
    that the re-association is numerically equivalent to the reference form.
    """

    def __init__(self, W):
        super().__init__(W)
        k = min(self.Nx, self.Nz)
        self.sigma = np.sqrt(self.spectrum[:k])
        self.Vk = self.V[:, :k]
        self.Uk = self.U[:, :k]

    def backward_mean(self, az, bz, ax, bx):
        t = self.Vk.T @ bz
        u = self.Uk.T @ bx
        sg = self.sigma if bz.ndim == 1 else self.sigma[:, None]
        D = az + ax * sg**2
        if self.Vk.shape[1] == self.Nz:     # Nx >= Nz: V_k spans everything
            return self.Vk @ ((t + sg * u) / D)
        return bz / az + self.Vk @ ((1 / D - 1 / az) * t + sg * u / D)

    def forward_mean(self, az, bz, ax, bx):
        t = self.Vk.T @ bz
        u = self.Uk.T @ bx
        sg = self.sigma if bz.ndim == 1 else self.sigma[:, None]
        D = az + ax * sg**2
        return self.Uk @ (sg * (t + sg * u) / D)


class LinearEconomyOracle(LinearThinOracle):
    """LinearThinOracle built from ONE economy SVD (no matrix_rank pass, no full V): the CPU baseline of
    bench.py at sizes where the reference's two full SVDs would take minutes.  Same formulas."""

    def __init__(self, W):
        W = np.asarray(W, dtype=float)
        self.W = W
        self.Nx, self.Nz = W.shape
        self.alpha = self.Nx / self.Nz
        U, s, VT = np.linalg.svd(W, full_matrices=False)
        self.rank = int(np.sum(s > s.max() * max(W.shape) * np.finfo(float).eps))     # np.linalg.matrix_rank
        spectrum = np.zeros(self.Nz)
        spectrum[:len(s)] = s**2
        self.spectrum = spectrum
        self.singular = spectrum[:self.rank]
        self.sigma = s
        self.Vk = VT.T
        self.Uk = U


# --------------------------------------------------------------------------
# Widening: circular convolution channel x = w * z, real signals
#   tramp/channels/linear/conv_channel.py:13-149 (real=True)
# --------------------------------------------------------------------------
class ConvOracle:
    def __init__(self, filt):
        filt = np.asarray(filt, dtype=np.float64)
        self.shape = filt.shape
        self.w_fft_bar = np.fft.fftn(filt)                       # :42-45
        self.w_fft = np.conjugate(self.w_fft_bar)
        self.spectrum = np.absolute(self.w_fft)**2
        self._axes = tuple(range(-filt.ndim, 0))                 # leading axes = ensemble

    def _fft(self, a):
        return np.fft.fftn(a, axes=self._axes)

    def _ifft(self, a):
        return np.real(np.fft.ifftn(a, axes=self._axes))

    def convolve(self, z):
        return self._ifft(self.w_fft * self._fft(z))             # :47-55

    def second_moment(self, tau_z):
        return tau_z * self.spectrum.mean()                      # :66-67

    def n_eff(self, az, ax):
        ":69-79"
        if ax == 0:
            return 0.
        if az / ax == 0:
            return 1.
        return np.mean(self.spectrum / (az / ax + self.spectrum))

    def backward_mean_fft(self, az, bz, ax, bx):
        resolvent = 1 / (az + ax * self.spectrum)                # :92
        return resolvent * (self._fft(bz) + self.w_fft_bar * self._fft(bx))

    def backward_mean(self, az, bz, ax, bx):
        return self._ifft(self.backward_mean_fft(az, bz, ax, bx))            # :81-103

    def forward_mean(self, az, bz, ax, bx):
        return self._ifft(self.w_fft * self.backward_mean_fft(az, bz, ax, bx))   # :105-109

    def backward_variance(self, az, ax):
        return (1 - self.n_eff(az, ax)) / az                     # :111-115

    def forward_variance(self, az, ax):
        if ax == 0:
            return np.mean(self.spectrum) / az                   # :117-120
        return self.n_eff(az, ax) / ax                           # :121-123

    def forward_message(self, az, bz, ax, bx):
        return ab_new(self.forward_mean(az, bz, ax, bx), self.forward_variance(az, ax), ax, bx)

    def backward_message(self, az, bz, ax, bx):
        return ab_new(self.backward_mean(az, bz, ax, bx), self.backward_variance(az, ax), az, bz)


class ComplexConvOracle(ConvOracle):
    """ConvChannel(filter, real=False) (conv_channel.py:25-32, 57-63, 83-103): x, w, z complex; every vector is the planar
    real array (2,) + shape (tramp/utils/misc.py:13-27), ensemble axes in front."""

    def __init__(self, filt):
        filt = np.asarray(filt, dtype=np.complex128)
        self.shape = filt.shape
        self.w_fft_bar = np.fft.fftn(filt)
        self.w_fft = np.conjugate(self.w_fft_bar)
        self.spectrum = np.absolute(self.w_fft)**2
        self._axes = tuple(range(-filt.ndim, 0))

    @staticmethod
    def _c(Z, nd):
        Z = np.asarray(Z)
        return np.take(Z, 0, axis=-nd - 1) + 1j * np.take(Z, 1, axis=-nd - 1)

    def _p(self, z):
        return np.stack([z.real, z.imag], axis=-len(self.shape) - 1)

    def _fft(self, Z):
        return np.fft.fftn(self._c(Z, len(self.shape)), axes=self._axes)

    def _ifft(self, a):
        return self._p(np.fft.ifftn(a, axes=self._axes))


class ComplexDFTOracle:
    "DFTChannel(real=False) (dft_channel.py:31-67): planar in, planar out, norm='ortho'"

    def __init__(self, ndim):
        self.nd = ndim
        self._axes = tuple(range(-ndim, 0))

    def forward_message(self, az, bz, ax, bx):
        z = ComplexConvOracle._c(bz, self.nd)
        x = np.fft.fftn(z, axes=self._axes, norm="ortho")
        return az, np.stack([x.real, x.imag], axis=-self.nd - 1)

    def backward_message(self, az, bz, ax, bx):
        x = ComplexConvOracle._c(bx, self.nd)
        z = np.fft.ifftn(x, axes=self._axes, norm="ortho")
        return ax, np.stack([z.real, z.imag], axis=-self.nd - 1)


def gaussian_filter(sigma, N):
    "tramp/utils/conv_filters.py:45-52"
    freq = np.fft.fftfreq(N)
    return np.real(np.fft.ifft(np.exp(-2 * (np.pi * sigma)**2 * freq**2)))


def stencil_filter(shape, first=None, second=None):
    """D1 . d/dx + D2 . d2/dx2 on a periodic grid (conv_filters.py:29-42,55-99):
    forward difference (-1, +1) at offsets (0, 1); second difference (-2, 1, 1) at (0, 1, -1)."""
    d = len(shape)
    f = np.zeros(shape)
    for ax in range(d):
        for coef, taps in ((first, {0: -1., 1: 1.}), (second, {0: -2., 1: 1., -1: 1.})):
            if coef is None or coef[ax] == 0:
                continue
            for off, w in taps.items():
                idx = [0] * d
                idx[ax] = off
                f[tuple(idx)] += coef[ax] * w
    return f


# --------------------------------------------------------------------------
# Widening: gradient channel x = grad z, x of shape (d,) + shape
#   tramp/channels/linear/gradient_channel.py:10-118 (real=True); conv_filters.py:102-119
# --------------------------------------------------------------------------
class GradientOracle:
    def __init__(self, shape):
        shape = tuple(shape)
        self.shape, self.d = shape, len(shape)
        self.filter = np.stack([stencil_filter(shape, first=[1. if a == ax else 0. for a in range(self.d)])
                                for ax in range(self.d)])
        self._axes = tuple(range(1, self.d + 1))
        self.w_fft_bar = np.fft.fftn(self.filter, axes=self._axes)            # :20-22
        self.w_fft = np.conjugate(self.w_fft_bar)
        self.spectrum = (np.absolute(self.w_fft)**2).sum(axis=0)              # :23

    def convolve(self, z):
        return np.real(np.fft.ifftn(self.w_fft * np.fft.fftn(z)[np.newaxis], axes=self._axes))   # :26-34

    def second_moment(self, tau_z):
        return tau_z * self.spectrum.mean() / self.d                          # :42-43

    def n_eff(self, az, ax):
        if ax == 0:
            return 0.
        if az / ax == 0:
            return 1.
        return np.mean(self.spectrum / (az / ax + self.spectrum))             # :45-55

    def backward_mean_fft(self, az, bz, ax, bx):
        bx_fft = np.fft.fftn(bx, axes=self._axes)                              # not over the direction axis
        resolvent = 1 / (az + ax * self.spectrum)
        return resolvent * (np.fft.fftn(bz) + (self.w_fft_bar * bx_fft).sum(axis=0))   # :57-68

    def backward_mean(self, az, bz, ax, bx):
        return np.real(np.fft.ifftn(self.backward_mean_fft(az, bz, ax, bx)))

    def forward_mean(self, az, bz, ax, bx):
        rz_fft = self.backward_mean_fft(az, bz, ax, bx)
        return np.real(np.fft.ifftn(self.w_fft * rz_fft[np.newaxis], axes=self._axes))   # :74-82

    def backward_variance(self, az, ax):
        return (1 - self.n_eff(az, ax)) / az                                   # :84-89

    def forward_variance(self, az, ax):
        if ax == 0:
            return np.mean(self.spectrum) / az                                 # :92-94
        return self.n_eff(az, ax) / (ax * self.d)                              # :95-97

    def forward_message(self, az, bz, ax, bx):
        return ab_new(self.forward_mean(az, bz, ax, bx), self.forward_variance(az, ax), ax, bx)

    def backward_message(self, az, bz, ax, bx):
        return ab_new(self.backward_mean(az, bz, ax, bx), self.backward_variance(az, ax), az, bz)


# --------------------------------------------------------------------------
# A14  Marchenko-Pastur closed-form linear channel (SE only)
#   tramp/channels/linear/analytical_linear_channel.py:23-53
#   tramp/ensembles/marchenko_pastur_ensemble.py:7-45
# --------------------------------------------------------------------------
class MarchenkoPasturOracle:
    def __init__(self, alpha):
        self.alpha = alpha
        self.z_max = (1 + np.sqrt(alpha))**2
        self.z_min = (1 - np.sqrt(alpha))**2

        def dens(z):
            return z * (np.sqrt((z - self.z_min) * (self.z_max - z)) / (2 * np.pi * z))
        # measure(lambda z: z): atomic part contributes f(0) = 0
        self.mean_spectrum = quad(dens, self.z_min, self.z_max)[0]

    def second_moment(self, tau_z):
        return tau_z * (self.mean_spectrum / self.alpha)

    def eta_transform(self, gamma):
        F = (np.sqrt(gamma * self.z_max + 1) - np.sqrt(gamma * self.z_min + 1))**2
        return 1 - F / (4 * gamma)

    def n_eff(self, az, ax):
        if ax == 0:
            return 0.
        if az / ax == 0:
            return min(1, self.alpha)
        return 1 - self.eta_transform(ax / az)

    def backward_error(self, az, ax, tau_z=None):
        az = np.maximum(1e-11, az)
        return (1 - self.n_eff(az, ax)) / az

    def forward_error(self, az, ax, tau_z=None):
        if ax == 0:
            return self.mean_spectrum / (self.alpha * az)
        return self.n_eff(az, ax) / (self.alpha * ax)


# --------------------------------------------------------------------------
# A11-A13  State-evolution errors (Gaussian quadrature of posterior variances)
#   tramp/utils/integration.py:13-73
# method="quadpack": the reference's adaptive QUADPACK calls (slow, faithful)
# method="gl": fixed Gauss-Legendre panels on [-10,10] (fast; what the HIP
#              kernels implement); agrees with quadpack to ~1e-9.
# --------------------------------------------------------------------------
def _gl_nodes(n_panels=40, order=16, lo=-10., hi=10.):
    x, w = np.polynomial.legendre.leggauss(order)
    edges = np.linspace(lo, hi, n_panels + 1)
    h = 0.5 * (edges[1:] - edges[:-1])
    c = 0.5 * (edges[1:] + edges[:-1])
    X = (c[:, None] + h[:, None] * x[None, :]).ravel()
    Wt = (h[:, None] * w[None, :]).ravel()
    return X, Wt


def gaussian_measure(m, s, f, method="quadpack"):
    "integration.py:13-28"
    if method == "quadpack":
        return quad(lambda x: norm_pdf(x) * f(m + s * x), -10, 10)[0]
    X, Wt = _gl_nodes(n_panels=1024, order=8)   # sharp features at large a need fine panels
    return np.sum(Wt * norm_pdf(X) * f(m + s * X))


def gaussian_measure_2d(m1, s1, m2, s2, f, method="quadpack"):
    "integration.py:31-47"
    if method == "quadpack":
        def integrand(x2, x1):
            return norm_pdf(x1) * norm_pdf(x2) * f(m1 + s1 * x1, m2 + s2 * x2)
        return dblquad(integrand, -10, 10, -10, 10)[0]
    X, Wt = _gl_nodes()
    X1, X2 = np.meshgrid(X, X, indexing="ij")
    W2 = (Wt * norm_pdf(X))[:, None] * (Wt * norm_pdf(X))[None, :]
    return np.sum(W2 * f(m1 + s1 * X1, m2 + s2 * X2))


def gaussian_measure_2d_full(mean, cov, f, method="quadpack"):
    "integration.py:50-73 (Cholesky-correlated)."
    L = np.linalg.cholesky(cov)
    if method == "quadpack":
        def integrand(x2, x1):
            y1, y2 = L @ [x1, x2] + mean
            return norm_pdf(x1) * norm_pdf(x2) * f(y1, y2)
        return dblquad(integrand, -10, 10, -10, 10)[0]
    X, Wt = _gl_nodes()
    X1, X2 = np.meshgrid(X, X, indexing="ij")
    W2 = (Wt * norm_pdf(X))[:, None] * (Wt * norm_pdf(X))[None, :]
    Y1 = L[0, 0] * X1 + L[0, 1] * X2 + mean[0]
    Y2 = L[1, 0] * X1 + L[1, 1] * X2 + mean[1]
    return np.sum(W2 * f(Y1, Y2))


def se_gb_prior_error(ax, rho=0.5, mean=0, var=1, method="quadpack"):
    """Prior.compute_forward_error (priors/base_prior.py:66-69) with
    GaussBernoulliPrior.beliefs_measure (gauss_bernoulli_prior.py:116-122)."""
    def v_func(bx):
        _, v = gb_prior_posterior(ax, np.asarray(bx, dtype=float),
                                  rho, mean, var, isotropic=False)
        return v
    mu0 = gaussian_measure(0, np.sqrt(ax), v_func, method)
    mu1 = gaussian_measure(ax * mean, np.sqrt(ax + (ax**2) * var), v_func, method)
    return (1 - rho) * mu0 + rho * mu1


def se_sgn_lik_error(az, tau_z, method="quadpack"):
    """Likelihood.compute_backward_error (base_likelihood.py:78-81) with
    SgnLikelihood.beliefs_measure (sgn_likelihood.py:79-92)."""
    mz_hat = az - 1 / tau_z
    assert mz_hat > 0, "az must be greater than 1/ tau_z"
    sz_eff = np.sqrt(mz_hat + (mz_hat**2) * tau_z)

    def branch(y):
        def g(bz):
            bz = np.asarray(bz, dtype=float)
            p = halfline_proba((y * bz) / az, 1 / az, +1)
            _, v, _ = halfline_moments((bz * y) / az, 1 / az, +1)
            return p * v
        return g
    return (gaussian_measure(0, sz_eff, branch(+1), method)
            + gaussian_measure(0, sz_eff, branch(-1), method))


def se_abs_lik_error(az, tau_z, method="quadpack"):
    "AbsLikelihood.beliefs_measure (abs_likelihood.py:60-69)."
    mz_hat = az - 1 / tau_z
    assert mz_hat > 0, "az must be greater than 1/tau_z"

    def integrand(z, xi_b):
        bz = mz_hat * z + np.sqrt(mz_hat) * xi_b
        y = np.abs(z)
        return (y**2) * (1 - np.tanh(bz * y)**2)
    return gaussian_measure_2d(0, np.sqrt(tau_z), 0, 1, integrand, method)


def se_pwl_channel_error(kind, direction, az, ax, tau_z, method="quadpack", param=None):
    """Channel.compute_forward/backward_error (channels/base_channel.py:29-41)
    with PiecewiseLinearChannel.beliefs_measure (piecewise_linear_channel.py:60-64)
    and LinearRegion.beliefs_measure (linear_region.py:83-101)."""
    def variance(bz, bx):
        _, v = pwl_channel_posterior(kind, direction, az, bz, ax, bx,
                                     isotropic=False, param=param)
        return v
    u_eff = np.maximum(0, az * tau_z - 1)
    total = 0.
    for side, slope, x0, zb in pwl_regions(kind, param):
        mean_x = ax * x0

        def integrand(bz, bx, side=side, slope=slope, x0=x0, zb=zb):
            return pwl_proba_ab(side, slope, x0, az, bz, ax, bx, zb) * variance(bz, bx)
        sx2 = ax * (slope**2 * ax * tau_z + 1)
        if ax == 0 or u_eff == 0 or slope == 0:
            total += gaussian_measure_2d(0, np.sqrt(az * u_eff), mean_x,
                                         np.sqrt(sx2), integrand, method)
        else:
            cov = np.array([[az * u_eff, slope * ax * u_eff],
                            [slope * ax * u_eff, sx2]])
            total += gaussian_measure_2d_full(np.array([0, mean_x]), cov,
                                              integrand, method)
    return total


def pwl_second_moment(kind, tau_z, param=None):
    "PiecewiseLinearChannel.second_moment (piecewise_linear_channel.py:21-25)."
    tot = 0.
    for side, slope, x0, zb in pwl_regions(kind, param):
        rz, vz, _ = halfline_moments(np.float64(0), np.float64(tau_z), side, zb)
        rx = slope * rz + x0
        vx = slope**2 * vz
        p = halfline_proba(np.float64(0), np.float64(tau_z), side, zb)
        tot += p * (rx**2 + vx)
    return tot


# --------------------------------------------------------------------------
# Row H  caller harness: chain-model EP / SE sweeps
#   tramp/algos/message_passing.py:108-127,187-269,330-357
#   tramp/base.py:152-170,329-413; tramp/variables/sub_variables.py:21-43
#   tramp/algos/callbacks.py:195-286; tramp/algos/initial_conditions.py:13-24
# --------------------------------------------------------------------------
class OracleFactor:
    """A factor of a chain model: callables working on (a, b) messages."""
    kind = "channel"       # 'prior' | 'channel' | 'likelihood'

    def forward(self, az, bz, ax, bx):
        raise NotImplementedError

    def backward(self, az, bz, ax, bx):
        raise NotImplementedError


class GBPriorFactor(OracleFactor):
    kind = "prior"

    def __init__(self, size, rho=0.5, mean=0, var=1):
        self.size, self.rho, self.mean, self.var = size, rho, mean, var

    def log_partition(self, az, bz, ax, bx):
        return gb_prior_log_partition(ax, bx, self.rho, self.mean, self.var)

    def forward(self, az, bz, ax, bx):
        return gb_prior_message(ax, bx, self.rho, self.mean, self.var)

    def forward_se(self, az, ax, tau_z, method="quadpack"):
        v = se_gb_prior_error(ax, self.rho, self.mean, self.var, method)
        return a_new(v, ax)

    def second_moment(self):
        return self.rho * (self.mean**2 + self.var)


def binary_prior_posterior(ax, bx, p_pos=0.5, isotropic=True):
    "tramp/priors/binary_prior.py:27,52-58; beliefs/binary.py:8-13"
    b = bx + 0.5 * np.log(p_pos / (1 - p_pos))
    v = 1 - np.tanh(b)**2
    return np.tanh(b), (v.mean() if isotropic else v)


def se_binary_prior_error(ax, p_pos=0.5, method="quadpack"):
    "binary_prior.py:82-86, base_prior.py:66-69"
    b0 = 0.5 * np.log(p_pos / (1 - p_pos))

    def v_func(bx):
        return 1 - np.tanh(bx + b0)**2
    return p_pos * gaussian_measure(+ax, np.sqrt(ax), v_func, method) + \
        (1 - p_pos) * gaussian_measure(-ax, np.sqrt(ax), v_func, method)


class BinaryPriorFactor(OracleFactor):
    kind = "prior"

    def __init__(self, size, p_pos=0.5):
        self.size, self.p_pos = size, p_pos

    def forward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(se_binary_prior_error(ax, self.p_pos, method), ax)

    def second_moment(self):
        return 1.

    def forward(self, az, bz, ax, bx):
        r, v = binary_prior_posterior(ax, bx, self.p_pos)
        return ab_new(r, v, ax, bx)


class L21PriorFactor(OracleFactor):
    kind = "prior"

    def __init__(self, size, gamma=1):
        self.size, self.gamma = size, gamma

    def forward(self, az, bz, ax, bx):
        return l21_prior_message(ax, bx, self.gamma)


class GaussPriorFactor(OracleFactor):
    kind = "prior"

    def __init__(self, size, mean=0, var=1):
        self.size, self.mean, self.var = size, mean, var

    def forward(self, az, bz, ax, bx):
        return gauss_prior_message(ax, bx, self.mean, self.var)

    def log_partition(self, az, bz, ax, bx):
        return gauss_prior_log_partition(ax, bx, self.mean, self.var)

    def forward_se(self, az, ax, tau_z, method="quadpack"):
        return 1 / self.var

    def second_moment(self):
        return self.mean**2 + self.var


class LinearFactor(OracleFactor):
    def __init__(self, W, thin=False):
        self.lin = (LinearEconomyOracle if thin == "economy" else LinearThinOracle if thin else LinearOracle)(W)

    def forward(self, az, bz, ax, bx):
        return self.lin.forward_message(az, bz, ax, bx)

    def backward(self, az, bz, ax, bx):
        return self.lin.backward_message(az, bz, ax, bx)

    def log_partition(self, az, bz, ax, bx):
        return linear_log_partition(self.lin, az, bz, ax, bx)

    def forward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(self.lin.forward_variance(az, ax), ax)

    def backward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(self.lin.backward_variance(az, ax), az)

    def second_moment(self, tau_z):
        return self.lin.second_moment(tau_z)


class ConvFactor(LinearFactor):
    def __init__(self, filt):
        self.lin = ConvOracle(filt)


class DFTFactor(OracleFactor):
    """tramp/channels/linear/dft_channel.py:46-64 (real z): messages pass through the unitary
    transform; complex vectors are planar (2,) + shape arrays."""

    def forward(self, az, bz, ax, bx):
        f = np.fft.fftn(bz, norm="ortho")
        return az, np.stack([f.real, f.imag])

    def backward(self, az, bz, ax, bx):
        return ax, np.real(np.fft.ifftn(bx[0] + 1j * bx[1], norm="ortho"))


class RotationFactor(OracleFactor):
    "tramp/channels/linear/rotation_channel.py:32-42"

    def __init__(self, R):
        self.R = np.asarray(R)

    def forward(self, az, bz, ax, bx):
        return az, self.R @ bz

    def backward(self, az, bz, ax, bx):
        return ax, self.R.T @ bx


class GradientFactor(LinearFactor):
    def __init__(self, shape):
        self.lin = GradientOracle(shape)


class UnitaryFactor(OracleFactor):
    "tramp/channels/linear/unitary_channel.py:58-72: planar complex (2, N) messages through U / U^H"

    def __init__(self, U):
        self.U = np.asarray(U)

    def forward(self, az, bz, ax, bx):
        f = self.U @ (bz[0] + 1j * bz[1])
        return az, np.stack([f.real, f.imag])

    def backward(self, az, bz, ax, bx):
        f = self.U.conj().T @ (bx[0] + 1j * bx[1])
        return ax, np.stack([f.real, f.imag])


class GaussChannelFactor(OracleFactor):
    "tramp/channels/noise/gaussian_channel.py:23-33: x = z + N(0, var) as a hidden factor"

    def __init__(self, var=1):
        self.a = 1 / var

    def forward(self, az, bz, ax, bx):
        kz = self.a / (self.a + az)
        return kz * az, kz * bz

    def backward(self, az, bz, ax, bx):
        kx = self.a / (self.a + ax)
        return kx * ax, kx * bx


class BiasFactor(OracleFactor):
    "tramp/channels/linear/bias_channel.py:21-29"

    def __init__(self, bias):
        self.bias = np.asarray(bias, dtype=float)

    def forward(self, az, bz, ax, bx):
        return az, bz + az * self.bias

    def backward(self, az, bz, ax, bx):
        return ax, bx - ax * self.bias

    def second_moment(self, tau_z):
        return tau_z + (self.bias**2).mean()


class SumFactor(OracleFactor):
    """tramp/channels/linear/sum_channel.py:26-43: x = sum_k z_k, K inputs.  forward(az, bz, ax, bx) takes lists
    az, bz; backward returns the K messages."""
    kind = "multi"

    def __init__(self, n_prev):
        self.n_prev = n_prev

    def forward(self, az, bz, ax, bx):
        v_bar = sum(1 / a for a in az)
        r_bar = sum(b / a for a, b in zip(az, bz))
        return 1 / v_bar, r_bar / v_bar

    def backward(self, az, bz, ax, bx):
        v_bar = sum(1 / a for a in az)
        r_bar = sum(b / a for a, b in zip(az, bz))
        vx, rx = 1 / ax, bx / ax
        vk = [vx + v_bar - 1 / a for a in az]
        rk = [rx - r_bar + b / a for a, b in zip(az, bz)]
        return [(1 / v, r / v) for v, r in zip(vk, rk)]

    def second_moment(self, *taus):
        return sum(taus)


class ReshapeFactor(OracleFactor):
    "tramp/channels/shape/reshape_channel.py:27-31"

    def __init__(self, prev_shape, next_shape):
        self.prev_shape, self.next_shape = prev_shape, next_shape

    def forward(self, az, bz, ax, bx):
        return az, bz.reshape(self.next_shape)

    def backward(self, az, bz, ax, bx):
        return ax, bx.reshape(self.prev_shape)


class MPFactor(OracleFactor):
    def __init__(self, alpha):
        self.mp = MarchenkoPasturOracle(alpha)

    def forward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(self.mp.forward_error(az, ax), ax)

    def backward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(self.mp.backward_error(az, ax), az)

    def second_moment(self, tau_z):
        return self.mp.second_moment(tau_z)


class PWLFactor(OracleFactor):
    def __init__(self, kind, param=None):
        self.kind_pwl = kind
        self.param = param

    def log_partition(self, az, bz, ax, bx):
        return pwl_channel_log_partition(self.kind_pwl, az, bz, ax, bx, self.param)

    def forward(self, az, bz, ax, bx):
        return pwl_channel_message(self.kind_pwl, "fwd", az, bz, ax, bx, self.param)

    def backward(self, az, bz, ax, bx):
        return pwl_channel_message(self.kind_pwl, "bwd", az, bz, ax, bx, self.param)

    def forward_se(self, az, ax, tau_z, method="quadpack"):
        v = se_pwl_channel_error(self.kind_pwl, "fwd", az, ax, tau_z, method, self.param)
        return a_new(v, ax)

    def backward_se(self, az, ax, tau_z, method="quadpack"):
        v = se_pwl_channel_error(self.kind_pwl, "bwd", az, ax, tau_z, method, self.param)
        return a_new(v, az)

    def second_moment(self, tau_z):
        return pwl_second_moment(self.kind_pwl, tau_z, self.param)


class GaussLikFactor(OracleFactor):
    kind = "likelihood"

    def __init__(self, y, var=1):
        self.y, self.var = y, var

    def log_partition(self, az, bz, ax, bx):
        return gauss_lik_log_partition(az, bz, self.y, self.var)

    def backward(self, az, bz, ax, bx):
        return gauss_lik_message(az, bz, self.y, self.var)

    def backward_se(self, az, ax, tau_z, method="quadpack"):
        return 1 / self.var


class SgnLikFactor(OracleFactor):
    kind = "likelihood"

    def __init__(self, y):
        self.y = y

    def log_partition(self, az, bz, ax, bx):
        return sgn_lik_log_partition(az, bz, self.y)

    def backward(self, az, bz, ax, bx):
        return sgn_lik_message(az, bz, self.y)

    def backward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(se_sgn_lik_error(az, tau_z, method), az)


class Pwl3Factor(OracleFactor):
    "three-region channels (HardTanh / HardSigmoid / SymmetricDoor) as chain factors"

    def __init__(self, kind, param=None):
        self.kind3, self.param = kind, param

    def forward(self, az, bz, ax, bx):
        return pwl3_channel_message(self.kind3, "fwd", az, bz, ax, bx, self.param)

    def backward(self, az, bz, ax, bx):
        return pwl3_channel_message(self.kind3, "bwd", az, bz, ax, bx, self.param)


class Pwl3LikFactor(OracleFactor):
    kind = "likelihood"

    def __init__(self, which, y, param=None):
        self.which, self.y, self.param = which, y, param

    def backward(self, az, bz, ax, bx):
        return pwl3_lik_message(self.which, az, bz, self.y, self.param)

    def backward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(se_pwl3_lik_error(self.which, az, tau_z, method, self.param), az)


class AbsLikFactor(OracleFactor):
    kind = "likelihood"

    def __init__(self, y):
        self.y = y

    def log_partition(self, az, bz, ax, bx):
        return abs_lik_log_partition(az, bz, self.y)

    def backward(self, az, bz, ax, bx):
        return abs_lik_message(az, bz, self.y)

    def backward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(se_abs_lik_error(az, tau_z, method), az)


# --------------------------------------------------------------------------
# log-partitions and the EP log-evidence
#   Factor.log_partition (tramp/base.py:367-375), Variable.compute_log_partition (:146-150),
#   MessagePassing.update_objective (tramp/algos/message_passing.py:306-328),
#   ExpectationPropagation.log_evidence (tramp/algos/expectation_propagation.py:21-27).
# NB the reference mixes conventions (restated as found): priors and likelihoods return the
# MEAN over components, channels and variables the SUM.
# --------------------------------------------------------------------------
def gb_prior_log_partition(ax, bx, rho=0.5, mean=0, var=1):
    "gauss_bernoulli_prior.py:79-83 with sparse.A (beliefs/sparse.py:5-6)"
    a0, b0, eta = gb_params(rho, mean, var)
    A = np.logaddexp(eta, normal_A(ax + a0, bx + b0)) - np.logaddexp(eta, normal_A(a0, b0))
    return A.mean()


def gauss_prior_log_partition(ax, bx, mean=0, var=1):
    "gaussian_prior.py:70-74"
    a0, b0 = 1 / var, mean / var
    return (normal_A(ax + a0, bx + b0) - normal_A(a0, b0)).mean()


def l1_prior_log_partition(ax, bx, gamma=1):
    "map_L1_norm_prior.py:73-76"
    rx, _ = l1_prior_posterior(ax, bx, gamma, isotropic=False)
    return (bx * rx - 0.5 * ax * rx**2 - gamma * np.abs(rx)).mean()


def gauss_lik_log_partition(az, bz, y, var=1):
    "gaussian_likelihood.py:51-56"
    ay = 1 / var
    by = ay * y
    return (normal_A(az + ay, bz + by) - normal_A(ay, by)).mean()


def sgn_lik_log_partition(az, bz, y):
    "sgn_likelihood.py:39-41 with positive.A (beliefs/positive.py:8-9)"
    b = bz * y
    return halfline_logZ(b / az, 1 / az, +1).mean()


def abs_lik_log_partition(az, bz, y):
    "abs_likelihood.py:38-40 with binary.A (beliefs/binary.py:4-5)"
    return (-0.5 * az * y**2 + np.logaddexp(bz * y, -bz * y)).mean()


def pwl_channel_log_partition(kind, az, bz, ax, bx, param=None):
    "piecewise_linear_channel.py:55-58: logsumexp of the regions' log-partitions, SUMMED"
    As = [pwl_region_terms(side, slope, x0, az, np.asarray(bz, float), ax, np.asarray(bx, float), zb)[2]
          for side, slope, x0, zb in pwl_regions(kind, param)]
    return np.logaddexp(As[0], As[1]).sum()


# --------------------------------------------------------------------------
# Region tables: the rest of the piecewise-linear family, region by region as the
# reference loops.  A region is (zmin, zmax, slope, x0); the tables are the reference constructors'
# (channels/activation/piecewise_linear_channel.py:72-132, likelihoods/piecewise_linear_likelihood.py:244-320).
# --------------------------------------------------------------------------
def f2_regions(name, likelihood=False, **kw):
    inf = np.inf
    if name == "relu":
        return [(0., inf, 1., 0.), (-inf, 0., 0., 0.)]
    if name == "l-relu":
        return [(0., inf, 1., 0.), (-inf, 0., float(kw["slope"]), 0.)]
    if name == "a-abs":
        s = float(kw.get("shift", 1e-4))
        return [(s, inf, 1., 0.), (-inf, s, -1., 0.)]
    if name == "h-tanh":
        return [(1., inf, 0., 1.), (-1., 1., 1., 0.), (-inf, -1., 0., -1.)]
    if name == "h-sigm":
        L = 3. if likelihood else 2.5          # the likelihood twin uses l = 3 (:302-309), the channel 2.5 (:117-123)
        return [(L, inf, 0., 1.), (-L, L, 1 / (2 * L), 0.5), (-inf, -L, 0., 0.)]
    if name == "door":
        w = float(kw["width"])
        return [(w, inf, 0., 1.), (-w, w, 0., -1.), (-inf, -w, 0., 1.)]
    raise KeyError(name)


def _scipy_logsumexp(As):
    "scipy.special.logsumexp(As, axis=0): the maximum is taken out unless it is not finite"
    A = np.stack(As)
    with np.errstate(all="ignore"):
        m = np.max(A, axis=0)
        m = np.where(np.isfinite(m), m, 0.)
        return np.log(np.sum(np.exp(A - m), axis=0)) + m


def region_trunc_logZ(zmin, zmax, r0, v0):
    "truncated_normal_logZ (tramp/utils/truncated_normal.py:294-298)"
    _, _, logp = _region_moments(zmin, zmax, r0, v0)
    return 0.5 * np.log(2 * np.pi * v0) + 0.5 * r0**2 / v0 + logp


def region_channel_log_partition(regions, az, bz, ax, bx):
    "LinearRegion.log_partitions (utils/linear_region.py:55-64), PiecewiseLinearChannel.compute_log_partition (:55-58)"
    bz = np.asarray(bz, dtype=float)
    bx = np.asarray(bx, dtype=float)
    As = []
    for zmin, zmax, slope, x0 in regions:
        a = az + slope**2 * ax
        b = bz + slope * (bx - ax * x0)
        As.append(region_trunc_logZ(zmin, zmax, b / a, 1 / a) - 0.5 * ax * x0**2 + bx * x0)
    return _scipy_logsumexp(As).sum()


def _region_lik_terms(regions, az, bz, y, abs_slope):
    """LinearRegionLikelihood.backward_mean / backward_variance / log_partitions (:31-61).  abs_slope=False is the
    reference as written (np.log(slope): NaN on a descending branch); True takes the Jacobian |slope|."""
    bz = np.asarray(bz, dtype=float)
    y = np.asarray(y, dtype=float) + np.zeros_like(bz)
    rs, vs, As = [], [], []
    for zmin, zmax, slope, x0 in regions:
        if slope == 0:
            contains = (y == x0)
            m, v, _ = _region_moments(zmin, zmax, bz / az, 1 / az)
            r_k, v_k = m, v
            A_k = region_trunc_logZ(zmin, zmax, bz / az, 1 / az)
        else:
            z = (y - x0) / slope
            contains = (zmin < z) & (z < zmax)
            r_k, v_k = z, np.zeros_like(z)
            with np.errstate(all="ignore"):
                A_k = -0.5 * az * (z**2) + bz * z - np.log(abs(slope) if abs_slope else slope)
        rs.append(np.where(contains, r_k, 0))
        vs.append(np.where(contains, v_k, 0))
        As.append(np.where(contains, A_k, -np.inf))
    return rs, vs, As


def region_lik_log_partition(regions, az, bz, y, abs_slope=False):
    "PiecewiseLinearLikelihood.compute_log_partition (:216-219)"
    _, _, As = _region_lik_terms(regions, az, bz, y, abs_slope)
    return _scipy_logsumexp(As).mean()


def region_lik_posterior(regions, az, bz, y, isotropic=True, abs_slope=False):
    "PiecewiseLinearLikelihood.compute_backward_posterior (:198-214): softmax of the regions' log-partitions"
    rs, vs, As = _region_lik_terms(regions, az, bz, y, abs_slope)
    A = np.stack(As)
    with np.errstate(all="ignore"):
        e = np.exp(A - A.max(axis=0, keepdims=True))
        ps = e / e.sum(axis=0, keepdims=True)
    r = sum(p_ * r_ for p_, r_ in zip(ps, rs))
    v = sum(p_ * v_ for p_, v_ in zip(ps, vs))
    n = len(rs)
    for i in range(n):
        for j in range(i + 1, n):
            v = v + ps[i] * ps[j] * (rs[i] - rs[j])**2
    if isotropic:
        v = v.mean()
    return r, v


def aabs_lik_message(az, bz, y, shift=1e-4, abs_slope=True):
    "AsymmetricAbsLikelihood (:260-265) through ab_new; abs_slope=False reproduces the reference's NaNs"
    r, v = region_lik_posterior(f2_regions("a-abs", shift=shift), az, bz, y, abs_slope=abs_slope)
    return ab_new(r, v, az, bz)


def se_aabs_lik_error(az, tau_z, shift=1e-4, nz=400, nxi=160):
    """beliefs_measure (:120-146, 210-214) for the two sloped regions of AsymmetricAbs: E over z ~ N(0, tau_z),
    xi ~ N(0, 1) of v(bz = mz z + sqrt(mz) xi, y = y(z)) with the |slope| Jacobian.  v jumps at |z| = |shift| (one
    compatible branch inside, two outside): Gauss-Legendre on the three segments, Gauss-Hermite in xi."""
    mz = az - 1 / tau_z
    assert mz > 0
    regions = f2_regions("a-abs", shift=shift)
    s0, s = np.sqrt(tau_z), abs(shift)
    xi, w = np.polynomial.hermite_e.hermegauss(nxi)
    g, gw = np.polynomial.legendre.leggauss(nz)
    total = 0.
    for lo, hi in ((-10 * s0, -s), (-s, s), (s, 10 * s0)):
        if hi <= lo:
            continue
        z = 0.5 * (hi - lo) * g + 0.5 * (hi + lo)
        y = np.where(z >= shift, z, -z)
        bz = mz * z[:, None] + np.sqrt(mz) * xi[None, :]
        _, v = region_lik_posterior(regions, az, bz, y[:, None], isotropic=False, abs_slope=True)
        f = (np.nan_to_num(v) @ (w / np.sqrt(2 * np.pi))) * np.exp(-0.5 * z**2 / tau_z) / np.sqrt(2 * np.pi * tau_z)
        total += 0.5 * (hi - lo) * float(gw @ f)
    return total


def variable_log_partition(a, b):
    "tramp/base.py:146-150"
    if a <= 0:
        return np.inf
    return 0.5 * np.sum(b**2 / a + np.log(2 * np.pi / a))


def linear_log_partition(lin, az, bz, ax, bx):
    "linear_channel.py:127-132 (lin: LinearOracle)"
    rz = lin.backward_mean(az, bz, ax, bx)
    b = bz + lin.W.T @ bx
    a = az + ax * lin.spectrum
    return 0.5 * np.sum(b * rz) + 0.5 * np.sum(np.log(2 * np.pi / a))


def chain_log_evidence(factors, fwd, bwd):
    """A_model of a chain f0 - x0 - f1 - ... (message_passing.py:306-328): sum of the node
    log-partitions minus the fwd-edge ones.  A SISO variable forwards its messages, so both of
    its edges carry the same (a, b) pair as the variable itself: nodes minus edges leaves
    sum_factors A_f - sum_variables A_x."""
    L = len(fwd)
    A = 0.
    for i, f in enumerate(factors):
        az, bz = fwd[i - 1] if i > 0 else (None, None)
        ax, bx = bwd[i] if i < L else (None, None)
        A += f.log_partition(az, bz, ax, bx)
    for i in range(L):
        A -= variable_log_partition(fwd[i][0] + bwd[i][0], fwd[i][1] + bwd[i][1])
    return A


def _rms(x):
    return np.sqrt(np.mean(x**2))       # callbacks.py:246-247


def ep_chain(factors, shapes, max_iter=200, damping=None, tol=1e-6,
             wait_increase=5, max_increase=0.2, record=None, init=None):
    """Expectation propagation on the chain  f0 - x0 - f1 - x1 - ... - fL.

    factors[0] is a prior, factors[-1] a likelihood, the others channels;
    variable i sits between factors[i] and factors[i+1]; ``shapes[i]`` is its
    shape.  Mirrors MessagePassing.iterate (message_passing.py:330-357) for a
    chain: ConstantInit(a=0, b=0) (initial_conditions.py:13-24); forward sweep
    in topological order, backward sweep in reverse, SISO variables copy
    messages through (sub_variables.py:21-33); constant damping only on
    factor->variable edges, on both a and b (message_passing.py:70-127);
    variable posterior r = sum b / sum a, v = 1 / sum a (base.py:152-161);
    EarlyStoppingEP (callbacks.py:250-286).

    ``record(it, fwd, bwd, rs, vs)`` is called after every iteration with the
    factor->variable messages and posteriors.  Returns dict with the final
    state and n_iter.  ``init(i, direction)`` -> (a, b) replaces the zero initial message of variable i's
    "fwd" / "bwd" edge (CustomInit, initial_conditions.py:45-85; message_passing.py:211-232).
    """
    L = len(factors) - 1            # number of variables
    # fwd[i] = message factor i -> variable i ; bwd[i] = factor i+1 -> variable i
    fwd = [(0, np.zeros(shapes[i])) for i in range(L)]
    bwd = [(0, np.zeros(shapes[i])) for i in range(L)]
    if init is not None:
        fwd = [(init(i, "fwd")[0], np.zeros(shapes[i]) + init(i, "fwd")[1]) for i in range(L)]
        bwd = [(init(i, "bwd")[0], np.zeros(shapes[i]) + init(i, "bwd")[1]) for i in range(L)]
    d = damping if damping else 0.
    old_rs = None
    old_state = None
    status = "max_iter"
    n_iter = 0
    rs = vs = None

    def check(a, b, where):
        if np.isnan(a):
            raise ValueError(f"{where} a is nan")    # message_passing.py:187-209
        if np.isnan(b).any():
            raise ValueError(f"{where} b is nan")

    def damp(old, new):
        if not d:
            return new
        return (d * old[0] + (1 - d) * new[0], d * old[1] + (1 - d) * new[1])

    for it in range(max_iter):
        # forward sweep
        for i in range(L):
            f = factors[i]
            if f.kind == "prior":
                new = f.forward(None, None, bwd[i][0], bwd[i][1])
            else:
                new = f.forward(fwd[i - 1][0], fwd[i - 1][1], bwd[i][0], bwd[i][1])
            check(new[0], new[1], f"f_{i}->x_{i}")
            fwd[i] = damp(fwd[i], new)
        # backward sweep
        for i in reversed(range(L)):
            f = factors[i + 1]
            if f.kind == "likelihood":
                new = f.backward(fwd[i][0], fwd[i][1], None, None)
            else:
                new = f.backward(fwd[i][0], fwd[i][1], bwd[i + 1][0], bwd[i + 1][1])
            check(new[0], new[1], f"f_{i+1}->x_{i}")
            bwd[i] = damp(bwd[i], new)
        # variable posteriors
        rs, vs = [], []
        for i in range(L):
            a_hat = fwd[i][0] + bwd[i][0]
            b_hat = fwd[i][1] + bwd[i][1]
            rs.append(b_hat / a_hat)
            vs.append(1. / a_hat)
        n_iter += 1
        if record is not None:
            record(it, fwd, bwd, rs, vs)
        if it == 0:
            old_rs = None
        if old_rs is not None:
            tols = [_rms(n - o) / _rms(n) for o, n in zip(old_rs, rs)]
            if max(tols) < tol:
                status = "converged"
                break
            if it > wait_increase and max(tols) > max_increase:
                # callbacks.py:277-283: "restoring old message dag" - messages AND posteriors go back to
                # the iteration before the increase; n_iter keeps counting the abandoned iteration
                fwd, bwd, rs, vs = old_state
                status = "diverged"
                break
        old_rs = rs
        old_state = (list(fwd), list(bwd), rs, vs)
    return dict(fwd=fwd, bwd=bwd, r=rs, v=vs, n_iter=n_iter, status=status)


def ep_chain_adaptive(factors, shapes, max_iter=10, record=None):
    """ep_chain with ``damping="adaptive"`` (tramp/algos/message_passing.py:108-117, 151-185): every edge of the message
    dag - the factor -> variable edges AND the variable -> factor edges a SISO variable passes its messages through
    (base.py:183-205) - searches beta = 1, 1/2 ... 1/512 for the first candidate old + beta (new - old) with
    dA = (A_target - A_edge)(candidate) - (A_target - A_edge)(old) >= 0, keeps the old message if none has, and takes the
    new one undamped in the first iteration.  A_target is the log-partition of the node the message goes INTO with all its
    incoming messages, A_edge that of the edge's variable with the two messages of the edge.

    Four messages per variable i between factors f = factors[i] and g = factors[i + 1]:
    ff (f -> x), vf (x -> g), fb (g -> x), vb (x -> f).  No stopping rule: exactly max_iter iterations.
    ``record(it, state)`` receives dict(ff, vf, fb, vb, r, v) after every iteration."""
    L = len(factors) - 1
    zero = lambda i: (0., np.zeros(shapes[i]))                            # noqa: E731  ConstantInit(a=0, b=0)
    ff, vf, fb, vb = ([zero(i) for i in range(L)] for _ in range(4))
    V = lambda m1, m2: variable_log_partition(m1[0] + m2[0], m1[1] + m2[1])      # noqa: E731

    def F(f, i_var, side, m):
        "log-partition of factor f with the message from variable i_var replaced by m"
        if f.kind == "prior":
            return f.log_partition(None, None, m[0], m[1])
        if f.kind == "likelihood":
            return f.log_partition(m[0], m[1], None, None)
        if side == "z":
            return f.log_partition(m[0], m[1], vb[i_var + 1][0], vb[i_var + 1][1])
        return f.log_partition(vf[i_var - 1][0], vf[i_var - 1][1], m[0], m[1])

    def damp(it, old, new, objective):
        if it == 0:
            return new                                                    # :152-153
        with np.errstate(all="ignore"):
            A_old = objective(old)
            for n in range(10):
                beta = 1 / 2**n
                cand = (old[0] + beta * (new[0] - old[0]), old[1] + beta * (new[1] - old[1]))
                if objective(cand) - A_old >= 0:
                    return cand
        return old

    for it in range(max_iter):
        for i in range(L):                                                # forward sweep: f_i, x_i, f_{i+1}, ...
            f, g = factors[i], factors[i + 1]
            new = f.forward(None, None, vb[i][0], vb[i][1]) if f.kind == "prior" else \
                f.forward(vf[i - 1][0], vf[i - 1][1], vb[i][0], vb[i][1])
            ff[i] = damp(it, ff[i], new, lambda m, i=i: V(m, fb[i]) - V(m, vb[i]))
            vf[i] = damp(it, vf[i], ff[i], lambda m, i=i, g=g: F(g, i, "z", m) - V(m, fb[i]))
        for i in reversed(range(L)):                                      # backward sweep
            f, g = factors[i], factors[i + 1]
            new = g.backward(vf[i][0], vf[i][1], None, None) if g.kind == "likelihood" else \
                g.backward(vf[i][0], vf[i][1], vb[i + 1][0], vb[i + 1][1])
            fb[i] = damp(it, fb[i], new, lambda m, i=i: V(m, ff[i]) - V(m, vf[i]))
            vb[i] = damp(it, vb[i], fb[i], lambda m, i=i, f=f: F(f, i, "x", m) - V(m, ff[i]))
        rs = [(ff[i][1] + fb[i][1]) / (ff[i][0] + fb[i][0]) for i in range(L)]
        vs = [1. / (ff[i][0] + fb[i][0]) for i in range(L)]
        if record is not None:
            record(it, dict(ff=list(ff), vf=list(vf), fb=list(fb), vb=list(vb), r=rs, v=vs))
    return dict(ff=ff, vf=vf, fb=fb, vb=vb, r=rs, v=vs)


def ep_glm_thin_batch(Uk, sigma, Vt, Y, rho, noise_var, damping=None, n_iter=10, rank=None):
    """ep_chain for the sparse GLM (GaussBernoulli prior - LinearThinOracle - Gaussian likelihood) on an ENSEMBLE
    of instances that share W = Uk diag(sigma) Vt: messages are [B, n] arrays, every scalar (a, v) a [B] array,
    the four matrix-vector products of an iteration become four GEMMs (Vt, Uk, Uk, Vt) - the host counterpart
    of what the HIP path runs per iteration on the headline shape (bench.py cpu_baseline "like_for_like").
    Same order of updates and same formulas as ep_chain + LinearThinOracle (pinned against them in
    tests/test_oracle_golden.py); no stopping rule: exactly n_iter iterations.  Returns dict(r, v, fwd, bwd)."""
    Uk, Vt, sigma = np.asarray(Uk), np.asarray(Vt), np.asarray(sigma, dtype=float)
    Y = np.atleast_2d(np.asarray(Y, dtype=float))
    B, M = Y.shape
    k, N = Vt.shape
    rank = int(np.sum(sigma > 0)) if rank is None else int(rank)
    sing2 = sigma[:rank]**2
    alpha = M / N
    d = damping if damping else 0.
    col = lambda a: np.asarray(a, dtype=float).reshape(B, 1)

    def n_eff(az, ax):                                                    # linear_channel.py:58-67, per instance
        out = np.zeros(B)
        for b in range(B):
            if ax[b] == 0:
                out[b] = 0.
            elif az[b] / ax[b] == 0:
                out[b] = rank / N
            else:
                out[b] = np.sum(sing2 / (az[b] / ax[b] + sing2)) / N
        return out

    def damp(old, new):
        if not d:
            return new
        return (d * old[0] + (1 - d) * new[0], d * old[1] + (1 - d) * new[1])

    fx = (np.zeros(B), np.zeros((B, N)))       # prior -> x
    bx = (np.zeros(B), np.zeros((B, N)))       # channel -> x
    fz = (np.zeros(B), np.zeros((B, M)))       # channel -> z
    bz = (np.zeros(B), np.zeros((B, M)))       # likelihood -> z
    for _ in range(n_iter):
        # forward sweep: prior, then the channel's forward message (linear_channel.py:85-89,99-105)
        r, v = [], []
        for b in range(B):
            rb, vb = gb_prior_posterior(bx[0][b], bx[1][b], rho)
            r.append(rb)
            v.append(vb)
        an, bn = ab_new(np.stack(r), col(v), col(bx[0]), bx[1])
        fx = damp(fx, (an.ravel(), bn))
        az, bzv, ax, bxv = fx[0], fx[1], bz[0], bz[1]
        t = bzv @ Vt.T
        u = bxv @ Uk
        D = col(az) + col(ax) * sigma[None, :]**2
        rx = (sigma[None, :] * (t + sigma[None, :] * u) / D) @ Uk.T
        ne = n_eff(az, ax)
        vx = np.where(ax == 0, np.mean(sing2) * rank / (M * np.where(az == 0, 1., az)),
                      ne / (alpha * np.where(ax == 0, 1., ax)))
        an, bn = ab_new(rx, col(vx), col(ax), bxv)
        fz = damp(fz, (an.ravel(), bn))
        # backward sweep: likelihood (constant message), then the channel's backward message (:69-83,91-97)
        bz = damp(bz, (np.full(B, 1. / noise_var), Y / noise_var))
        az, bzv, ax, bxv = fx[0], fx[1], bz[0], bz[1]
        t = bzv @ Vt.T
        u = bxv @ Uk
        D = col(az) + col(ax) * sigma[None, :]**2
        if k == N:
            rz = ((t + sigma[None, :] * u) / D) @ Vt
        else:
            rz = bzv / col(az) + ((1 / D - 1 / col(az)) * t + sigma[None, :] * u / D) @ Vt
        vz = (1 - n_eff(np.maximum(1e-11, az), ax)) / np.maximum(1e-11, az)
        an, bn = ab_new(rz, col(vz), col(az), bzv)
        bx = damp(bx, (an.ravel(), bn))
    ahx, ahz = fx[0] + bx[0], fz[0] + bz[0]
    return dict(r=[(fx[1] + bx[1]) / col(ahx), (fz[1] + bz[1]) / col(ahz)], v=[1. / ahx, 1. / ahz],
                fwd=[fx, fz], bwd=[bx, bz])


def ep_tree(factors, variables, edges, shapes, max_iter=200, damping=None, tol=1e-6,
            wait_increase=5, max_increase=0.2, record=None, a0=0., b0=0., init=None):
    """Expectation propagation on a tree-structured model (message_passing.py:211-269,330-357 with
    the variable nodes of tramp/base.py:152-211).

    ``factors``: {name: OracleFactor}; ``variables``: names; ``edges``: directed (src, dst) name pairs
    of the model DAG (factor -> variable or variable -> factor); ``shapes``: {variable: shape}.
    Both message kinds are stored: a factor reads the variable -> factor messages its neighbours
    sent when the sweep last VISITED them.  Forward sweep: nodes in topological order, a factor sends
    to its successor, a variable to its successors (sum of all incoming minus the target's own);
    backward sweep: reverse order, towards predecessors.  Constant damping on factor -> variable
    messages only (message_passing.py:70-127).  ``record(it, msg, rs, vs)`` after each iteration.
    """
    nodes = list(factors) + list(variables)
    pred = {n: [s for s, t in edges if t == n] for n in nodes}
    succ = {n: [t for s, t in edges if s == n] for n in nodes}
    order, indeg = [], {n: len(pred[n]) for n in nodes}
    ready = [n for n in nodes if indeg[n] == 0]
    while ready:
        n = ready.pop(0)
        order.append(n)
        for t in succ[n]:
            indeg[t] -= 1
            if indeg[t] == 0:
                ready.append(t)
    assert len(order) == len(nodes), "model graph has a cycle"
    msg = {}
    for s, t in edges:
        v = s if s in shapes else t
        msg[(s, t)] = (a0, b0 * np.ones(shapes[v]))               # ConstantInit (initial_conditions.py:13-24)
        msg[(t, s)] = (a0, b0 * np.ones(shapes[v]))
        if init is not None:                                      # init(variable, direction) -> (a, b), e.g. NoisyInit
            msg[(s, t)] = init(v, "fwd")
            msg[(t, s)] = init(v, "bwd")
    d = damping if damping else 0.

    def damp(old, new):
        return new if not d else (d * old[0] + (1 - d) * new[0], d * old[1] + (1 - d) * new[1])

    def factor_send(f, direction):
        fac = factors[f]
        if fac.kind == "multi":                                  # K inputs, one output (SumChannel)
            nxt = succ[f][0]
            az, bz = [msg[(p, f)][0] for p in pred[f]], [msg[(p, f)][1] for p in pred[f]]
            with np.errstate(all="ignore"):
                if direction == "fwd":
                    news, tgts = [fac.forward(az, bz, *msg[(nxt, f)])], [nxt]
                else:
                    news, tgts = fac.backward(az, bz, *msg[(nxt, f)]), pred[f]
            for new, t in zip(news, tgts):
                msg[(f, t)] = damp(msg[(f, t)], new)
            return
        prev = pred[f][0] if pred[f] else None
        nxt = succ[f][0] if succ[f] else None
        target = nxt if direction == "fwd" else prev
        if target is None:
            return
        if fac.kind == "prior":
            new = fac.forward(None, None, *msg[(nxt, f)])
        elif fac.kind == "likelihood":
            new = fac.backward(*msg[(prev, f)], None, None)
        elif direction == "fwd":
            new = fac.forward(*msg[(prev, f)], *msg[(nxt, f)])
        else:
            new = fac.backward(*msg[(prev, f)], *msg[(nxt, f)])
        if np.isnan(new[0]) or np.isnan(new[1]).any():
            raise ValueError(f"{f}->{target} message is nan")
        msg[(f, target)] = damp(msg[(f, target)], new)

    def variable_send(v, targets):
        inc = [msg[(f, v)] for f in pred[v] + succ[v]]
        a_hat = sum(m[0] for m in inc)                                          # base.py:152-155
        b_hat = sum(m[1] for m in inc)
        for f in targets:
            msg[(v, f)] = (a_hat - msg[(f, v)][0], b_hat - msg[(f, v)][1])      # base.py:186-211

    old_rs, n_iter, rs, vs = None, 0, None, None
    old_state, status = None, "max_iter"
    for it in range(max_iter):
        for n in order:
            if n in factors:
                factor_send(n, "fwd")
            else:
                variable_send(n, succ[n])
        for n in reversed(order):
            if n in factors:
                factor_send(n, "bwd")
            else:
                variable_send(n, pred[n])
        rs, vs = {}, {}
        for v in variables:
            inc = [msg[(f, v)] for f in pred[v] + succ[v]]
            a_hat = sum(m[0] for m in inc)
            rs[v] = sum(m[1] for m in inc) / a_hat
            vs[v] = 1. / a_hat
        n_iter += 1
        if record is not None:
            record(it, msg, rs, vs)
        if old_rs is not None:
            tols = [_rms(rs[v] - old_rs[v]) / _rms(rs[v]) for v in variables]
            if max(tols) < tol:
                status = "converged"
                break
            if it > wait_increase and max(tols) > max_increase:
                # callbacks.py:277-283 "restoring old message dag": messages and posteriors of the iteration before
                # the increase (every message is replaced, never mutated, so a shallow copy of the dict holds them)
                old_msg, rs, vs = old_state
                msg.clear()
                msg.update(old_msg)
                status = "diverged"
                break
        old_rs = rs
        old_state = (dict(msg), rs, vs)
    return dict(msg=msg, r=rs, v=vs, n_iter=n_iter, status=status)


def se_chain(factors, max_iter=200, damping=None, tol=1e-6, a0=0.,
             method="quadpack", record=None):
    """State evolution on the same chain, messages are scalars ``a``.

    StateEvolution (algos/state_evolution.py:5-27); second moments propagated
    as Model.init_second_moments (models/base_model.py:111-124); stopping rule
    EarlyStopping on v (callbacks.py:195-243).
    """
    L = len(factors) - 1
    taus = []
    for i in range(L):
        f = factors[i]
        taus.append(f.second_moment() if f.kind == "prior"
                    else f.second_moment(taus[i - 1]))
    fwd = [a0] * L
    bwd = [a0] * L
    d = damping if damping else 0.
    old_vs = None
    n_iter = 0
    vs = None
    for it in range(max_iter):
        for i in range(L):
            f = factors[i]
            if f.kind == "prior":
                new = f.forward_se(None, bwd[i], None, method)
            else:
                new = f.forward_se(fwd[i - 1], bwd[i], taus[i - 1], method)
            if np.isnan(new):
                raise ValueError(f"f_{i}->x_{i} a is nan")
            fwd[i] = d * fwd[i] + (1 - d) * new if d else new
        for i in reversed(range(L)):
            f = factors[i + 1]
            if f.kind == "likelihood":
                new = f.backward_se(fwd[i], None, taus[i], method)
            else:
                new = f.backward_se(fwd[i], bwd[i + 1], taus[i], method)
            if np.isnan(new):
                raise ValueError(f"f_{i+1}->x_{i} a is nan")
            bwd[i] = d * bwd[i] + (1 - d) * new if d else new
        vs = [1. / (fwd[i] + bwd[i]) for i in range(L)]
        n_iter += 1
        if record is not None:
            record(it, list(fwd), list(bwd), vs)
        if it == 0:
            old_vs = None
        if old_vs is not None:
            if max(abs(o - n) for o, n in zip(old_vs, vs)) < tol:
                break
            if it > 5 and max(n - o for o, n in zip(old_vs, vs)) > 0.2:
                break
        old_vs = vs
    return dict(fwd=fwd, bwd=bwd, v=vs, tau=taus, n_iter=n_iter)


def se_tree(factors, variables, edges, max_iter=200, damping=None, tol=1e-6, a0=0., method="quadpack",
            record=None):
    """State evolution on a tree-structured model: the scalar counterpart of ``ep_tree`` (StateEvolution,
    algos/state_evolution.py:5-27, with Variable.forward_state_evolution / backward_state_evolution,
    base.py:213-233).  ``record(it, msg, vs)`` after each iteration; msg[(src, dst)] is the scalar a."""
    nodes = list(factors) + list(variables)
    pred = {n: [s for s, t in edges if t == n] for n in nodes}
    succ = {n: [t for s, t in edges if s == n] for n in nodes}
    order, indeg = [], {n: len(pred[n]) for n in nodes}
    ready = [n for n in nodes if indeg[n] == 0]
    while ready:
        n = ready.pop(0)
        order.append(n)
        for t in succ[n]:
            indeg[t] -= 1
            if indeg[t] == 0:
                ready.append(t)
    taus = {}
    for n in order:                                               # base_model.py:111-124
        if n in factors and succ[n]:
            tau = factors[n].second_moment(*[taus[p] for p in pred[n]])
            for t in succ[n]:
                taus[t] = tau
    msg = {}
    for s, t in edges:
        msg[(s, t)] = a0
        msg[(t, s)] = a0
    d = damping if damping else 0.

    def factor_send(f, direction):
        fac = factors[f]
        if fac.kind == "multi":                                  # SumChannel (sum_channel.py:45-56)
            nxt = succ[f][0]
            az = [msg[(p, f)] for p in pred[f]]
            v_bar = sum(1 / a for a in az)
            if direction == "fwd":
                news, tgts = [1 / v_bar], [nxt]
            else:
                news, tgts = [1 / (1 / msg[(nxt, f)] + v_bar - 1 / a) for a in az], pred[f]
            for new, t in zip(news, tgts):
                msg[(f, t)] = d * msg[(f, t)] + (1 - d) * new if d else new
            return
        prev = pred[f][0] if pred[f] else None
        nxt = succ[f][0] if succ[f] else None
        target = nxt if direction == "fwd" else prev
        if target is None:
            return
        if fac.kind == "prior":
            new = fac.forward_se(None, msg[(nxt, f)], None, method)
        elif fac.kind == "likelihood":
            new = fac.backward_se(msg[(prev, f)], None, taus[prev], method)
        elif direction == "fwd":
            new = fac.forward_se(msg[(prev, f)], msg[(nxt, f)], taus[prev], method)
        else:
            new = fac.backward_se(msg[(prev, f)], msg[(nxt, f)], taus[prev], method)
        if np.isnan(new):
            raise ValueError(f"{f}->{target} a is nan")
        msg[(f, target)] = d * msg[(f, target)] + (1 - d) * new if d else new

    def variable_send(v, targets):
        a_hat = sum(msg[(f, v)] for f in pred[v] + succ[v])
        for f in targets:
            msg[(v, f)] = a_hat - msg[(f, v)]

    old_vs, n_iter, vs = None, 0, None
    for it in range(max_iter):
        for n in order:
            if n in factors:
                factor_send(n, "fwd")
            else:
                variable_send(n, succ[n])
        for n in reversed(order):
            if n in factors:
                factor_send(n, "bwd")
            else:
                variable_send(n, pred[n])
        vs = {v: 1. / sum(msg[(f, v)] for f in pred[v] + succ[v]) for v in variables}
        n_iter += 1
        if record is not None:
            record(it, dict(msg), dict(vs))
        if old_vs is not None:
            if max(abs(old_vs[v] - vs[v]) for v in variables) < tol:
                break
            if it > 5 and max(vs[v] - old_vs[v] for v in variables) > 0.2:
                break
        old_vs = vs
    return dict(msg=msg, v=vs, tau=taus, n_iter=n_iter)


# --------------------------------------------------------------------------
# Synthetic teacher data (legacy RandomState is version-stable)
#   tramp/ensembles/gaussian_ensemble.py:11-21
#   tramp/priors/gauss_bernoulli_prior.py:38-42
# --------------------------------------------------------------------------
def gaussian_ensemble(rs, M, N):
    return rs.randn(M, N) / np.sqrt(N)


def gb_sample(rs, N, rho, mean=0., var=1.):
    g = mean + np.sqrt(var) * rs.standard_normal(N)
    return g * rs.binomial(n=1, size=N, p=rho)


# ---- the int8-digit engine of the dense passes (tramp_amd/csrc/gemm_sliced.h): an integer restatement -------------
# Not an algorithm of the reference (its products are numpy's `V @ x`, tramp/channels/linear/linear_channel.py:85-118):
# this restates the DEVICE arithmetic - digit split, exact integer accumulation per split and weight, the order of the
# f64 operations that combine them - so that the GPU result can be compared BIT FOR BIT (integer work), and the result
# itself against the exact product.
def digit_scale(amax):
    """per-row power-of-two scale 2^(ilogb(max) + 1): 1 for a zero row, never below 2^-960 (the slicing's multiplier
    2^62 / scale must stay finite), NaN for a non-finite row and for a maximum of 2^1023 or more (scale 2^1024 = inf)"""
    amax = np.asarray(amax, dtype=np.float64)
    sc = np.ones_like(amax)
    fin = np.isfinite(amax) & (amax < 2.0 ** 1023)
    pos = fin & (amax > 0)
    m, e = np.frexp(np.where(pos, amax, 1.0))           # amax = m 2^e, m in [0.5, 1): ilogb + 1 = e
    sc = np.where(pos, np.ldexp(1.0, np.maximum(e, -960)), sc)
    return np.where(fin, sc, np.nan)


def digit_split(a, S):
    """a [rows, k] -> (digits int64 [S, rows, k] in [-128, 127] (|digit 0| <= 65), scale [rows]) with
    a ~ scale sum_i digits[i] 2^-(6 + 8 i): N = rint(a 2^(6 + 8 (S - 1)) / scale), balanced base-256 digits of N."""
    a = np.asarray(a, dtype=np.float64)
    sc = digit_scale(np.max(np.abs(a), axis=1) if a.shape[1] else np.zeros(a.shape[0]))
    inv = np.where(np.isnan(sc), 0.0, 2.0 ** (6 + 8 * (S - 1)) / np.where(np.isnan(sc), 1.0, sc))
    with np.errstate(invalid="ignore"):
        N = np.rint(np.where(np.isfinite(a), a, 0.0) * inv[:, None]).astype(np.int64)
    dig = np.empty((S,) + a.shape, dtype=np.int64)
    for i in range(S - 1, 0, -1):
        d = ((N & 0xff) ^ 0x80) - 0x80                  # the low byte as a signed one
        dig[i] = d
        N = (N - d) >> 8
    dig[0] = N
    return dig, sc


def digit_matvec(mat, x, S=7):
    """out[b, r] = sum_c mat[r, c] x[b, c] as the device computes it (gemm_sliced.hip:run_sliced + gemm_sliced_kernel +
    mfma_fold_kernel, mode 0): k blocks of 64, splits = the smallest multiple of 8 that keeps a split below
    2^31 / (S 2^14) contraction indices (at most the number of k blocks), per split and weight d = i + j < S the exact
    integer sum, s = sum_{d = S-1 .. 0} (double) acc_d 2^-(12 + 8 d), partial = (s row_scale) x_scale, splits added in
    order."""
    mat, x = np.asarray(mat, dtype=np.float64), np.asarray(x, dtype=np.float64)
    R, k = mat.shape
    A, rs = digit_split(mat, S)
    X, xs = digit_split(x, S)
    nkb = (k + 63) // 64
    kmax = (131072 // S) // 64
    nsplit = 8
    while (nkb + nsplit - 1) // nsplit > kmax:
        nsplit += 8
    nsplit = min(nsplit, nkb)
    kps = (nkb + nsplit - 1) // nsplit
    out = np.zeros((x.shape[0], R))
    for sp in range(nsplit):
        c0, c1 = sp * kps * 64, min((sp + 1) * kps * 64, k)
        s = np.zeros((x.shape[0], R))
        for d in range(S - 1, -1, -1):
            acc = np.zeros((x.shape[0], R), dtype=np.int64)
            if c1 > c0:
                for i in range(d + 1):
                    acc += X[d - i][:, c0:c1] @ A[i][:, c0:c1].T
            assert np.all(np.abs(acc) < 2 ** 31)        # what the device holds in an int32
            s = s + acc.astype(np.float64) * 2.0 ** -(12 + 8 * d)
        with np.errstate(invalid="ignore"):
            out = out + (s * rs[None, :]) * xs[:, None]
    return out


# --------------------------------------------------------------------------
# Round 6: the rest of the reference's separable priors (Exponential, Positive, GaussianMixture) and the
# complex-valued factors of phase retrieval (ModulusLikelihood, ComplexLinearChannel).
# Pinned by tests/golden/priors2.npz and tests/golden/complex.npz (tools/make_golden_r6.py).
# --------------------------------------------------------------------------
def positive_rv(a, b):
    "beliefs/positive.py:12-17: mean and variance of N(b/a, 1/a) restricted to [0, inf)"
    m, v, _ = halfline_moments(b / a, 1 / a, +1)
    return m, v


def positive_A(a, b):
    "beliefs/positive.py:8-9 (truncated_normal_logZ on [0, inf))"
    return halfline_logZ(b / a, 1 / a, +1)


def exp_prior_posterior(ax, bx, mean=1, isotropic=True):
    "exponential_prior.py:25-26 (natural parameter b0 = -1/mean), :54-60"
    r, v = positive_rv(ax, np.asarray(bx, float) - 1. / mean)
    return r, (v.mean() if isotropic else v)


def exp_prior_message(ax, bx, mean=1):
    r, v = exp_prior_posterior(ax, bx, mean)
    return ab_new(r, v, ax, bx)


def exp_prior_log_partition(ax, bx, mean=1):
    "exponential_prior.py:62-65 with exponential.A(b) = -log(-b) (beliefs/exponential.py:4-5)"
    b0 = -1. / mean
    return (positive_A(ax, np.asarray(bx, float) + b0) + np.log(-b0)).mean()


def positive_prior_posterior(ax, bx, isotropic=True):
    "positive_prior.py:23-25 (a0 = 1, b0 = 0), :56-63"
    r, v = positive_rv(ax + 1., np.asarray(bx, float) + 0.)
    return r, (v.mean() if isotropic else v)


def positive_prior_message(ax, bx):
    r, v = positive_prior_posterior(ax, bx)
    return ab_new(r, v, ax, bx)


def positive_prior_log_partition(ax, bx):
    "positive_prior.py:65-69"
    return (positive_A(ax + 1., np.asarray(bx, float)) - positive_A(1., 0.)).mean()


def gm_params(probs, means, vars_):
    "gaussian_mixture_prior.py:36-39: a_k = 1/var_k, b_k = mean_k/var_k, eta_k = log p_k - normal.A(a_k, b_k)"
    probs, means, vars_ = (np.asarray(t, float) for t in (probs, means, vars_))
    a0, b0 = 1 / vars_, means / vars_
    return a0, b0, np.log(probs) - normal_A(a0, b0)


def _gm_terms(ax, bx, probs, means, vars_):
    a0, b0, eta = gm_params(probs, means, vars_)
    bx = np.asarray(bx, float)
    a = ax + a0.reshape((-1,) + (1,) * bx.ndim)                 # (K, ...) as :89-91
    b = bx + b0.reshape((-1,) + (1,) * bx.ndim)
    xi = eta.reshape((-1,) + (1,) * bx.ndim) + normal_A(a, b)
    return a, b, xi


def gm_prior_posterior(ax, bx, probs=(0.5, 0.5), means=(-1, 1), vars_=(1, 1), isotropic=True):
    "gaussian_mixture_prior.py:88-97 with beliefs/mixture.py:23-41"
    from scipy.special import softmax
    a, b, xi = _gm_terms(ax, bx, probs, means, vars_)
    s = softmax(xi, axis=0)
    rk = b / a
    r = (s * rk).sum(axis=0)
    Dr = 0.5 * sum(sk * sl * (r_k - r_l)**2 for sk, r_k in zip(s, rk) for sl, r_l in zip(s, rk))
    v = Dr + (s / a).sum(axis=0)
    return r, (v.mean() if isotropic else v)


def gm_prior_message(ax, bx, probs=(0.5, 0.5), means=(-1, 1), vars_=(1, 1)):
    r, v = gm_prior_posterior(ax, bx, probs, means, vars_)
    return ab_new(r, v, ax, bx)


def gm_prior_log_partition(ax, bx, probs=(0.5, 0.5), means=(-1, 1), vars_=(1, 1)):
    "gaussian_mixture_prior.py:99-104 with mixture.A (beliefs/mixture.py:18-20)"
    from scipy.special import logsumexp
    a0, b0, eta = gm_params(probs, means, vars_)
    _, _, xi = _gm_terms(ax, bx, probs, means, vars_)
    return (logsumexp(xi, axis=0) - logsumexp(eta + normal_A(a0, b0))).mean()


def se_gm_prior_error(ax, probs=(0.5, 0.5), means=(-1, 1), vars_=(1, 1), method="quadpack"):
    "gaussian_mixture_prior.py:132-137 (beliefs_measure) through base_prior.py:66-69"
    def v_func(bx):
        return gm_prior_posterior(ax, np.asarray(bx, float), probs, means, vars_, isotropic=False)[1]
    return sum(p * gaussian_measure(ax * m, np.sqrt(ax + ax**2 * v), v_func, method)
               for p, m, v in zip(probs, means, vars_))


def gm_second_moment(probs, means, vars_):
    "gaussian_mixture_prior.py:53-58"
    return sum(p * (m**2 + v) for p, m, v in zip(probs, means, vars_))


class _SeparablePriorFactor(OracleFactor):
    kind = "prior"

    def forward(self, az, bz, ax, bx):
        r, v = self.posterior(ax, bx)
        return ab_new(r, v, ax, bx)


class ExpPriorFactor(_SeparablePriorFactor):
    def __init__(self, size, mean=1):
        self.size, self.mean = size, mean

    def posterior(self, ax, bx):
        return exp_prior_posterior(ax, bx, self.mean)

    def log_partition(self, az, bz, ax, bx):
        return exp_prior_log_partition(ax, bx, self.mean)

    def second_moment(self):
        return 2 * self.mean**2


class PositivePriorFactor(_SeparablePriorFactor):
    def __init__(self, size):
        self.size = size

    def posterior(self, ax, bx):
        return positive_prior_posterior(ax, bx)

    def log_partition(self, az, bz, ax, bx):
        return positive_prior_log_partition(ax, bx)

    def second_moment(self):
        return 1.


class GMPriorFactor(_SeparablePriorFactor):
    def __init__(self, size, probs=(0.5, 0.5), means=(-1, 1), vars_=(1, 1)):
        self.size, self.probs, self.means, self.vars = size, probs, means, vars_

    def posterior(self, ax, bx):
        return gm_prior_posterior(ax, bx, self.probs, self.means, self.vars)

    def log_partition(self, az, bz, ax, bx):
        return gm_prior_log_partition(ax, bx, self.probs, self.means, self.vars)

    def forward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(se_gm_prior_error(ax, self.probs, self.means, self.vars, method), ax)

    def second_moment(self):
        return gm_second_moment(self.probs, self.means, self.vars)


# ---- complex variables as planar real arrays Z[0] = Re z, Z[1] = Im z (tramp/utils/misc.py:13-27) ----
def ive_ratio(r):
    "modulus_likelihood.py:9-20: I1(r)/I0(r) from the exponentially scaled Bessel functions, 1 beyond 1e9"
    from scipy.special import ive
    r = np.asarray(r, float)
    out = np.ones_like(r)
    ind = r < 1e9
    out[ind] = ive(1, r[ind]) / ive(0, r[ind])
    return out


def modulus_lik_posterior(az, bz, y, isotropic=True):
    """modulus_likelihood.py:82-93: bz planar (2, ...), y = |z| observed;
    rz = (bz/|bz|) y I(|bz| y), vz = mean of y^2 (1 - I^2) / 2 (the 1/2 averages the two real coordinates)."""
    bz = np.asarray(bz, float)
    c = bz[0] + 1j * bz[1]
    b = np.abs(c)
    Ir = ive_ratio(b * y)
    nz = b != 0
    unit = np.zeros_like(c)
    unit[nz] = c[nz] / b[nz]
    rz = unit * y * Ir
    vz = 0.5 * y**2 * (1 - Ir**2)
    return np.stack([rz.real, rz.imag]), (vz.mean() if isotropic else vz)


def modulus_lik_message(az, bz, y):
    r, v = modulus_lik_posterior(az, bz, y)
    return ab_new(r, v, az, np.asarray(bz, float))


def modulus_lik_log_partition(az, bz, y):
    "modulus_likelihood.py:125-129 (np.absolute of the PLANAR array, elementwise, as found; mean / 2)"
    from scipy.special import ive
    b = np.abs(np.asarray(bz, float))
    A = -0.5 * az * y**2 + np.log(2 * np.pi * y * ive(0, b * y)) + b * y
    return A.mean() / 2


def se_modulus_lik_error(az, tau_z, method="quadpack"):
    """ModulusLikelihood.beliefs_measure (modulus_likelihood.py:101-120) through base_likelihood.py:78-81:
    u_eff = max(0, az tau_z - 1); u_eff = 0: a 1-D average of sqrt(2 pi az) relu(y) v(0, y), y = xi / sqrt(az);
    otherwise the 2-D average of (2 pi / sqrt(u_eff)) relu(b) relu(y) ive(0, b y) v(b, y) with b = sqrt(az u_eff) xi_b,
    y = b / az + xi_y / sqrt(az), v(b, y) = y^2 (1 - I(b y)^2) / 2."""
    from scipy.special import ive

    def v_func(b, y):
        Ir = ive_ratio(np.atleast_1d(np.abs(b) * y))[0]
        return 0.5 * y**2 * (1 - Ir**2)
    u_eff = np.maximum(0, az * tau_z - 1)
    if u_eff == 0:
        def f_y(xi_y):
            y = xi_y / np.sqrt(az)
            return np.sqrt(2 * np.pi * az) * np.maximum(0, y) * v_func(0., y)
        return gaussian_measure(0, 1, f_y, method)
    sz_eff = np.sqrt(az * u_eff)

    def f_scaled(xi_b, xi_y):
        b = sz_eff * xi_b
        y = b / az + xi_y / np.sqrt(az)
        return (2 * np.pi / np.sqrt(u_eff)) * np.maximum(0, b) * np.maximum(0, y) * ive(0, b * y) * v_func(b, y)
    return gaussian_measure_2d(0, 1, 0, 1, f_scaled, method)


class ModulusLikFactor(OracleFactor):
    kind = "likelihood"

    def __init__(self, y):
        self.y = None if y is None else np.asarray(y, float)

    def backward(self, az, bz, ax, bx):
        return modulus_lik_message(az, bz, self.y)

    def backward_se(self, az, ax, tau_z, method="quadpack"):
        return a_new(se_modulus_lik_error(az, tau_z, method), az)


class ComplexLinearOracle:
    """ComplexLinearChannel (tramp/channels/linear/complex_linear_channel.py:17-162): x = W z for complex W, messages
    planar (2, N).  Restated on the complex SVD as the reference."""

    def __init__(self, W):
        W = np.asarray(W)
        self.W = W
        self.Nx, self.Nz = W.shape
        self.rank = np.linalg.matrix_rank(W)
        self.alpha = self.Nx / self.Nz
        U, s, VH = np.linalg.svd(W, full_matrices=True)
        self.U, self.V = U, VH.conj().T
        self.S = np.zeros((self.Nx, self.Nz))
        self.S[:len(s), :len(s)] = np.diag(s)
        self.spectrum = np.diag(self.S.conj().T @ self.S)
        self.singular = self.spectrum[:self.rank]

    @staticmethod
    def _c(Z):
        Z = np.asarray(Z, float)
        return Z[0] + 1j * Z[1]

    @staticmethod
    def _p(z):
        return np.stack([z.real, z.imag])

    def n_eff(self, az, ax):
        if ax == 0:
            return 0.
        if az / ax == 0:
            return self.rank / self.Nz
        return np.sum(self.singular / (az / ax + self.singular)) / self.Nz

    def backward_mean(self, az, bz, ax, bx):
        bz, bx = self._c(bz), self._c(bx)
        resolvent = 1 / (az + ax * self.spectrum)
        rz = self.V @ (resolvent * (self.V.conj().T @ bz + self.S.conj().T @ (self.U.conj().T @ bx)))
        return self._p(rz)

    def forward_mean(self, az, bz, ax, bx):
        return self._p(self.W @ self._c(self.backward_mean(az, bz, ax, bx)))

    def backward_variance(self, az, ax):
        return (1 - self.n_eff(az, ax)) / az

    def forward_variance(self, az, ax):
        if ax == 0:
            return np.mean(self.singular) * self.rank / (self.Nx * az)
        return self.n_eff(az, ax) / (self.alpha * ax)

    def backward_message(self, az, bz, ax, bx):
        return ab_new(self.backward_mean(az, bz, ax, bx), self.backward_variance(az, ax), az, np.asarray(bz, float))

    def forward_message(self, az, bz, ax, bx):
        return ab_new(self.forward_mean(az, bz, ax, bx), self.forward_variance(az, ax), ax, np.asarray(bx, float))

    def log_partition(self, az, bz, ax, bx):
        "complex_linear_channel.py:143-150"
        rz = self.backward_mean(az, bz, ax, bx)
        b = self._p(self._c(bz) + self.W.conj().T @ self._c(bx))
        return 0.5 * np.sum(b * rz) + np.sum(np.log(2 * np.pi / (az + ax * self.spectrum)))


class ComplexLinearFactor(OracleFactor):
    def __init__(self, W):
        self.lin = ComplexLinearOracle(W)

    def forward(self, az, bz, ax, bx):
        return self.lin.forward_message(az, bz, ax, bx)

    def backward(self, az, bz, ax, bx):
        return self.lin.backward_message(az, bz, ax, bx)


# ------------------------------------------------------------------------------------------------------------------
# DuplicateChannel x_1 = ... = x_K = z (tramp/channels/shape/duplicate_channel.py:24-31; the messages through
# base_channel.py:102-105 and base.py:250-255, 329-365).  Test infrastructure like the rest of this file.
def duplicate_posterior(az, bz, ax, bx):
    a = az + sum(ax)
    b = bz + sum(bx)
    return b / a, 1.0 / a


def duplicate_backward_message(az, bz, ax, bx):
    r, v = duplicate_posterior(az, bz, ax, bx)
    return ab_new(r, v, az, bz)


def duplicate_forward_message(az, bz, ax, bx):
    "one message per output: compute_ab_new(rz, vz, ax_k, bx_k)"
    r, v = duplicate_posterior(az, bz, ax, bx)
    outs = [ab_new(r, v, a, b) for a, b in zip(ax, bx)]
    return [o[0] for o in outs], [o[1] for o in outs]
