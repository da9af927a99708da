"""Moment / log-partition consistency of the device factors, the way the reference's own unit tests
check its modules (tramp/tests/test_priors.py:57-69 `test_prior_grad_EP_scalar`,
tramp/tests/test_likelihoods.py:86-92, tramp/checks/check_gradients.py): the posterior a factor
returns is the gradient of its own log-partition,

    r = dA/db,   v = d2A/db2,   tau = r^2 + v = -2 dA/da      (MAP priors: q = r^2 = -2 dA/da),

here with both sides computed on the GPU through the C ABI (central differences of
`compute_log_partition` against `compute_*_posterior`).  One-component factors, one (a, b) point
per ensemble instance, so a whole grid is a single launch.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H = 1e-4            # central-difference step: truncation ~ h^2, rounding ~ eps / h^2
TOL = 2e-6


def grid(seed, n_pts, b_scale=3.0):
    rs = np.random.RandomState(seed)
    a = rs.uniform(0.3, 4.0, n_pts)
    b = rs.uniform(-b_scale, b_scale, (n_pts, 1))
    return a, b


def d1(f, x, h):
    return (f(x + h) - f(x - h)) / (2 * h)


def d2(f, x, h):
    return (f(x + h) - 2 * f(x) + f(x - h)) / h**2


def prior_cases(T, ctx):
    kw = dict(size=1, ctx=ctx)
    return [("gb", T.GaussBernoulliPrior(rho=0.3, mean=0.4, var=1.3, **kw), False),
            ("gauss", T.GaussianPrior(mean=-0.2, var=0.7, **kw), False),
            ("l1", T.MAP_L1NormPrior(gamma=0.8, **kw), True),
            # round 6 (and the binary prior of round 4): the same identities for the rest of the separable priors
            ("binary", T.BinaryPrior(p_pos=0.3, **kw), False),
            ("exponential", T.ExponentialPrior(mean=0.7, **kw), False),
            ("positive", T.PositivePrior(**kw), False),
            ("mixture", T.GaussianMixturePrior(probs=[0.2, 0.5, 0.3], means=[-2.0, 0.0, 1.5], vars=[0.3, 0.05, 2.0], **kw), False)]


def test_prior_posterior_is_gradient_of_log_partition(ctx):
    import tramp_amd as T
    a, b = grid(1, 257)
    for name, prior, is_map in prior_cases(T, ctx):
        if is_map:                                   # keep clear of the soft-threshold kinks |b| = gamma
            b = np.where(np.abs(np.abs(b) - 0.8) < 0.05, b + 0.2 * np.sign(b), b)
        r, v = prior.compute_forward_posterior(a, b)
        A_b = lambda bb: np.asarray(prior.compute_log_partition(a, bb))            # noqa: E731
        A_a = lambda aa: np.asarray(prior.compute_log_partition(aa, b))            # noqa: E731
        np.testing.assert_allclose(d1(A_b, b, H), r[:, 0], rtol=2e-6, atol=TOL, err_msg=name + ": r = dA/db")
        np.testing.assert_allclose(d2(A_b, b, H), v, rtol=2e-6, atol=20 * TOL, err_msg=name + ": v = d2A/db2")
        second = r[:, 0] ** 2 if is_map else r[:, 0] ** 2 + v
        np.testing.assert_allclose(-2 * d1(A_a, a, H), second, rtol=2e-6, atol=TOL, err_msg=name + ": -2 dA/da")


def test_binary_prior_log_partition_value(ctx):
    "binary_prior.py:64-67 literally: mean(logaddexp(b, -b)) - logaddexp(b0, -b0) - ax / 2 with b = bx + b0, also where exp(b) overflows"
    import tramp_amd as T
    rs = np.random.RandomState(5)
    bx = np.concatenate([rs.standard_normal(500) * 3, [0.0, 800.0, -800.0, 1e-300]])
    for p_pos, ax in ((0.3, 1.7), (0.5, 0.2), (0.999, 40.0)):
        pr = T.BinaryPrior(size=bx.size, p_pos=p_pos, ctx=ctx)
        b0 = 0.5 * np.log(p_pos / (1 - p_pos))
        want = np.mean(np.logaddexp(bx + b0, -(bx + b0)) - np.logaddexp(b0, -b0) - 0.5 * ax)
        assert pr.compute_log_partition(ax, bx) == pytest.approx(want, rel=1e-13)


def test_likelihood_posterior_is_gradient_of_log_partition(ctx):
    import tramp_amd as T
    a, b = grid(2, 193)
    rs = np.random.RandomState(3)
    y_g = rs.randn(a.size, 1)
    y_s = np.sign(rs.randn(a.size, 1))
    y_a = np.abs(rs.randn(a.size, 1)) + 0.1
    z = rs.randn(a.size, 1) * 1.5
    y_relu = np.maximum(z, 0.0)                                   # half of the observations sit on the atom y = 0
    y_leaky = np.where(z > 0, z, 0.1 * z)
    y_tanh = np.clip(z, -1.0, 1.0)                                # atoms at +-1
    y_sigm = np.clip(z / 5.0 + 0.5, 0.0, 1.0)                     # atoms at 0 and 1 (rare at this scale)
    y_door = np.where(np.abs(z) < 0.8, -1.0, 1.0)
    for name, lik, y in (("gaussian", T.GaussianLikelihood(y=y_g, var=0.4, ctx=ctx), y_g),
                         ("sgn", T.SgnLikelihood(y=y_s, ctx=ctx), y_s),
                         ("abs", T.AbsLikelihood(y=y_a, ctx=ctx), y_a),
                         # the rest of the piecewise-linear family (round 4 / 5 kernels): the identities per observation,
                         # on the point-mass branches (r = the pre-image, v = 0) and on the truncated-normal ones
                         ("relu", T.ReluLikelihood(y=y_relu, ctx=ctx), y_relu),
                         ("l-relu", T.LeakyReluLikelihood(slope=0.1, y=y_leaky, ctx=ctx), y_leaky),
                         ("h-tanh", T.HardTanhLikelihood(y=y_tanh, ctx=ctx), y_tanh),
                         ("h-sigm", T.HardSigmoidLikelihood(y=y_sigm, ctx=ctx), y_sigm),
                         ("door", T.SymmetricDoorLikelihood(width=0.8, y=y_door, ctx=ctx), y_door)):
        r, v = lik.compute_backward_posterior(a, b, y)
        A_b = lambda bb: np.asarray(lik.compute_log_partition(a, bb, y))            # noqa: E731
        A_a = lambda aa: np.asarray(lik.compute_log_partition(aa, b, y))            # noqa: E731
        np.testing.assert_allclose(d1(A_b, b, H), r[:, 0], rtol=2e-6, atol=TOL, err_msg=name + ": r = dA/db")
        np.testing.assert_allclose(d2(A_b, b, H), v, rtol=2e-6, atol=20 * TOL, err_msg=name + ": v = d2A/db2")
        np.testing.assert_allclose(-2 * d1(A_a, a, H), r[:, 0] ** 2 + v, rtol=2e-6, atol=TOL,
                                   err_msg=name + ": -2 dA/da")


@pytest.mark.parametrize("name", ["sgn", "abs", "relu"])
def test_channel_posteriors_are_gradients_of_log_partition(ctx, name):
    import tramp_amd as T
    ch = {"sgn": T.SgnChannel, "abs": T.AbsChannel, "relu": T.ReluChannel}[name](ctx=ctx)
    az, bz = grid(4, 161, 2.0)
    ax, bx = grid(5, 161, 2.0)
    rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
    rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
    A = lambda az_, bz_, ax_, bx_: np.asarray(ch.compute_log_partition(az_, bz_, ax_, bx_))      # noqa: E731
    np.testing.assert_allclose(d1(lambda t: A(az, t, ax, bx), bz, H), rz[:, 0], rtol=2e-6, atol=TOL)
    np.testing.assert_allclose(d1(lambda t: A(az, bz, ax, t), bx, H), rx[:, 0], rtol=2e-6, atol=TOL)
    np.testing.assert_allclose(d2(lambda t: A(az, t, ax, bx), bz, H), vz, rtol=2e-6, atol=20 * TOL)
    np.testing.assert_allclose(d2(lambda t: A(az, bz, ax, t), bx, H), vx, rtol=2e-6, atol=20 * TOL)
    np.testing.assert_allclose(-2 * d1(lambda t: A(t, bz, ax, bx), az, H), rz[:, 0] ** 2 + vz, rtol=2e-6, atol=TOL)
    np.testing.assert_allclose(-2 * d1(lambda t: A(az, bz, t, bx), ax, H), rx[:, 0] ** 2 + vx, rtol=2e-6, atol=TOL)


# tramp/tests/test_channels.py:53-110: channel posteriors against explicit numerical integration of
# p(z) ~ N(z | az, bz) N(f(z) | ax, bx), on the reference's own three records
RECORDS = [dict(az=2.1, bz=2.0, ax=2.0, bx=2.0), dict(az=2.0, bz=+1.6, ax=1.5, bx=1.3),
           dict(az=2.0, bz=-1.6, ax=1.5, bx=1.3)]


def explicit_moments(f, breaks, az, bz, ax, bx):
    from scipy.integrate import quad

    def w(z):
        x = f(z)
        return np.exp(-0.5 * az * z * z + bz * z - 0.5 * ax * x * x + bx * x)

    def integral(g):
        pts = [-30.0] + list(breaks) + [30.0]
        return sum(quad(lambda z: g(z) * w(z), lo, hi, epsabs=0, epsrel=1e-13, limit=200)[0]
                   for lo, hi in zip(pts[:-1], pts[1:]))
    Z = integral(lambda z: 1.0)
    rz = integral(lambda z: z) / Z
    rx = integral(f) / Z
    vz = integral(lambda z: (z - rz) ** 2) / Z
    vx = integral(lambda z: (f(z) - rx) ** 2) / Z
    return rz, vz, rx, vx


def test_channel_posteriors_match_explicit_integrals(ctx):
    import tramp_amd as T
    cases = [(T.AbsChannel(ctx=ctx), np.abs, [0.0]),
             (T.SgnChannel(ctx=ctx), np.sign, [0.0]),
             (T.ReluChannel(ctx=ctx), lambda z: max(z, 0.0), [0.0]),
             (T.LeakyReluChannel(slope=0.1, ctx=ctx), lambda z: z if z > 0 else 0.1 * z, [0.0]),
             (T.HardTanhChannel(ctx=ctx), lambda z: min(max(z, -1.0), 1.0), [-1.0, 1.0])]
    for ch, f, breaks in cases:
        for rec in RECORDS:
            az, bz, ax, bx = rec["az"], rec["bz"], rec["ax"], rec["bx"]
            rz, vz, rx, vx = explicit_moments(f, breaks, az, bz, ax, bx)
            one = lambda t: np.array([t])                                              # noqa: E731
            rx_hat, vx_hat = ch.compute_forward_posterior(az, one(bz), ax, one(bx))
            rz_hat, vz_hat = ch.compute_backward_posterior(az, one(bz), ax, one(bx))
            msg = f"{type(ch).__name__} {rec}"
            np.testing.assert_allclose([rz_hat[0], vz_hat, rx_hat[0], vx_hat], [rz, vz, rx, vx], rtol=1e-9,
                                       atol=1e-11, err_msg=msg)
