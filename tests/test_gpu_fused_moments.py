"""GPU parity of the producer-fused prior moments (csrc/prior_hint.h): the kernel that writes the
message into a GaussBernoulli / L1 prior also leaves sum v_i, and the prior update then reads that
message once.  Checked here: (1) the fused path is actually taken (hint statistics), (2) its result
equals the prior's own two-pass result to reduction-order rounding and the oracle / reference goldens,
for every producer: the dense channel's GEMV fold (batch <= 2), MFMA fold (batch >= 3), rotated-basis
epilogue, and the inverse FFT of the convolution channel.

Reference: tramp/priors/gauss_bernoulli_prior.py:70-77, map_L1_norm_prior.py:66-71 (posteriors),
tramp/base.py:250-255 (message update), tramp/channels/linear/linear_channel.py:69-117.
"""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _own_fft_engine(monkeypatch):
    """These tests exercise the own FFT engine at sizes below the default switch-over (csrc/conv.hip own_engine_wanted:
    rocFFT under 32768 points per signal until round 5; now the default anyway); tests that compare engines set TRAMP_CONV_FFT themselves."""
    monkeypatch.setenv("TRAMP_CONV_FFT", "own")


@pytest.fixture(scope="module")
def T():
    import tramp_amd
    return tramp_amd


def _tol(dtype):
    return dict(rtol=1e-11, atol=1e-12) if dtype == np.float64 else dict(rtol=1e-3, atol=1e-4)


def _priors(T, n, dtype, ctx):
    return [("gb", T.GaussBernoulliPrior(size=n, rho=0.15, mean=0.1, var=1.3, dtype=dtype, ctx=ctx)),
            ("l1", T.MAP_L1NormPrior(size=n, gamma=0.7, dtype=dtype, ctx=ctx))]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("batch", [1, 2, 5, 64, 70])
@pytest.mark.parametrize("damping", [0.0, 0.3])
def test_dense_producer_leaves_prior_moments(T, ctx, dtype, batch, damping):
    rs = np.random.RandomState(batch)
    M, N = 40, 9216          # longer than one workgroup's register tile (4096 f64 / 8192 f32): the streaming kernels run
    W = rs.randn(M, N) / np.sqrt(N)
    ch = T.LinearChannel(W, dtype=dtype, ctx=ctx, svd="host")
    az = ctx.array(rs.uniform(0.5, 3, batch)); ax = ctx.array(rs.uniform(0.2, 5, batch))
    bz = ctx.array(rs.standard_normal((batch, N)), dtype); bx = ctx.array(rs.standard_normal((batch, M)), dtype)
    for name, prior in _priors(T, N, dtype, ctx):
        kind = prior.moment_hint()
        a_msg = ctx.array(rs.uniform(0.5, 2, batch))                      # old message (for damping)
        b_msg = ctx.array(rs.standard_normal((batch, N)), dtype)
        kw = dict(out_a=a_msg, out_b=b_msg)
        if damping:
            kw.update(damping=damping, old_a=a_msg, old_b=b_msg)
        armed0, used0 = ctx.prior_hint_stats()
        ctx._call("tramp_prior_hint", kind[0], kind[1], kind[2], kind[3], N, batch, b_msg.ptr, a_msg.ptr)
        ch.compute_backward_message(az, bz, ax, bx, **kw)
        a1, b1 = ctx.empty((batch,)), ctx.empty((batch, N), dtype)
        prior.compute_forward_message(a_msg, b_msg, out_a=a1, out_b=b1)    # fused: reads b_msg once
        armed1, used1 = ctx.prior_hint_stats()
        assert (armed1 - armed0, used1 - used0) == (1, 1), (name, batch)
        a2, b2 = ctx.empty((batch,)), ctx.empty((batch, N), dtype)
        prior.compute_forward_message(a_msg, b_msg, out_a=a2, out_b=b2)    # hint consumed: the prior's own pass 1
        assert ctx.prior_hint_stats() == (armed1, used1)
        t = _tol(dtype)
        np.testing.assert_allclose(a1.get(), a2.get(), rtol=max(t["rtol"], 1e-12))
        np.testing.assert_allclose(b1.get(), b2.get(), rtol=t["rtol"], atol=t["atol"] * np.abs(b2.get()).max())
        if dtype == np.float64:                                            # and the oracle on the same message
            am, bm = a_msg.get(), b_msg.get()
            for i in range(min(batch, 3)):
                if name == "gb":
                    a_ref, b_ref = orc.gb_prior_message(am[i], bm[i], rho=0.15, mean=0.1, var=1.3)
                else:
                    a_ref, b_ref = orc.l1_prior_message(am[i], bm[i], gamma=0.7)
                np.testing.assert_allclose(a1.get()[i], a_ref, rtol=1e-10)
                np.testing.assert_allclose(b1.get()[i], b_ref, rtol=1e-9, atol=1e-10 * np.abs(b_ref).max())


def test_hint_is_ignored_when_it_does_not_match(T, ctx):
    "another buffer, another shape, or a prior with other parameters: the prior runs its own first pass"
    rs = np.random.RandomState(0)
    M, N, B = 24, 4608, 4
    ch = T.LinearChannel(rs.randn(M, N) / 8, ctx=ctx, svd="host")
    az = ctx.array(rs.uniform(1, 2, B)); ax = ctx.array(rs.uniform(1, 2, B))
    bz = ctx.array(rs.standard_normal((B, N))); bx = ctx.array(rs.standard_normal((B, M)))
    a_msg, b_msg = ctx.empty((B,)), ctx.empty((B, N))
    other = ctx.empty((B, N))
    p = T.GaussBernoulliPrior(size=N, rho=0.2, ctx=ctx)
    p2 = T.GaussBernoulliPrior(size=N, rho=0.3, ctx=ctx)
    k = p.moment_hint()
    # (a) hint names a different buffer: producer does not arm
    armed0, used0 = ctx.prior_hint_stats()
    ctx._call("tramp_prior_hint", k[0], k[1], k[2], k[3], N, B, other.ptr, a_msg.ptr)
    ch.compute_backward_message(az, bz, ax, bx, out_a=a_msg, out_b=b_msg)
    assert ctx.prior_hint_stats() == (armed0, used0)
    # (b) armed, but consumed by a prior with other parameters: falls back, result is that prior's own
    ctx._call("tramp_prior_hint", k[0], k[1], k[2], k[3], N, B, b_msg.ptr, a_msg.ptr)
    ch.compute_backward_message(az, bz, ax, bx, out_a=a_msg, out_b=b_msg)
    a_f, b_f = p2.compute_forward_message(a_msg.get(), b_msg.get())
    a1, b1 = ctx.empty((B,)), ctx.empty((B, N))
    p2.compute_forward_message(a_msg, b_msg, out_a=a1, out_b=b1)
    assert ctx.prior_hint_stats() == (armed0 + 1, used0)
    np.testing.assert_allclose(a1.get(), a_f, rtol=1e-13)
    np.testing.assert_allclose(b1.get(), b_f, rtol=1e-12, atol=1e-13)
    # (c) and the matching prior afterwards finds the hint consumed
    p.compute_forward_message(a_msg, b_msg, out_a=a1, out_b=b1)
    assert ctx.prior_hint_stats() == (armed0 + 1, used0)


@pytest.mark.parametrize("batch", [1, 3, 64])
@pytest.mark.parametrize("damping", [None, 0.1])
def test_ep_takes_the_fused_path_and_matches_the_oracle(T, ctx, batch, damping):
    "sparse-GLM EP: every iteration after the first has a single-pass prior update; results as the oracle's EP"
    rs = np.random.RandomState(3)
    N, M, rho, nv = 4352, 192, 0.05, 1e-2          # N past the single-workgroup tile (4096 f64)
    W = orc.gaussian_ensemble(rs, M, N)
    X = np.stack([orc.gb_sample(rs, N, rho) for _ in range(batch)])
    Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((batch, M))
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.LinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    armed0, used0 = ctx.prior_hint_stats()
    n_it = 12
    ep.iterate(max_iter=n_it, callback=T.PassCallback(), damping=damping)
    armed1, used1 = ctx.prior_hint_stats()
    assert armed1 - armed0 == n_it and used1 - used0 == n_it - 1
    data = ep.get_variables_data()["x"]
    r = np.atleast_2d(data["r"]); v = np.atleast_1d(data["v"])
    lin = orc.LinearFactor(W)
    for b in range(min(batch, 3)):
        res = orc.ep_chain([orc.GBPriorFactor(N, rho=rho), lin, orc.GaussLikFactor(Y[b], nv)], [N, M],
                           max_iter=n_it, damping=damping, tol=0.0, max_increase=np.inf)
        np.testing.assert_allclose(r[b], res["r"][0], rtol=1e-7, atol=1e-8 * np.abs(res["r"][0]).max())
        np.testing.assert_allclose(v[b], res["v"][0], rtol=1e-8)


def test_conv_channel_inverse_fft_leaves_prior_moments(T, ctx):
    "the own FFT engine's fused message update is a producer too (n = 2 * 2^a 5^b)"
    rs = np.random.RandomState(5)
    n, B, rho, nv = 16384, 3, 0.1, 1e-3
    ch = T.ConvChannel(rs.randn(n) / np.sqrt(n), ctx=ctx)
    X = rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n))
    Y = ch.convolve(ctx.array(X)).get() + np.sqrt(nv) * rs.standard_normal((B, n))
    model = (T.GaussBernoulliPrior(size=n, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ ch
             @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    armed0, used0 = ctx.prior_hint_stats()
    ep.iterate(max_iter=8, callback=T.PassCallback(), damping=0.1)
    armed1, used1 = ctx.prior_hint_stats()
    if armed1 == armed0:
        pytest.skip("this length runs on rocFFT (no fused message update)")
    assert armed1 - armed0 == 8 and used1 - used0 == 7
    r = ep.get_variables_data()["x"]["r"]
    fac = orc.ConvFactor(np.asarray(ch.filter))
    for b in range(B):
        res = orc.ep_chain([orc.GBPriorFactor(n, rho=rho), fac, orc.GaussLikFactor(Y[b], nv)], [n, n],
                           max_iter=8, damping=0.1, tol=0.0, max_increase=np.inf)
        np.testing.assert_allclose(r[b], res["r"][0], rtol=1e-6, atol=1e-8 * np.abs(res["r"][0]).max())


@pytest.mark.parametrize("shape", [(8192,), (128, 128), (16, 64, 32)], ids=lambda s: "x".join(map(str, s)))
def test_block_pass_epilogue_leaves_prior_moments(T, ctx, shape):
    """The one-block-pass inverse (short 1-D signals) and the last pass of the N-D plans are producers as well: the
    Gauss-Bernoulli prior behind a 1-D / 2-D / 3-D convolution channel runs single-pass, same EP as the oracle.
    (Up to 4096 components the prior is one workgroup that ignores the hint: the 1-D case is just above.)"""
    rs = np.random.RandomState(9)
    n, B, rho, nv = int(np.prod(shape)), 3, 0.1, 1e-3
    f = rs.standard_normal(shape) / np.sqrt(n)
    ch = T.ConvChannel(f, ctx=ctx)
    X = rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n))
    Y = ch.convolve(ctx.array(X)).get() + np.sqrt(nv) * rs.standard_normal((B, n))
    model = (T.GaussBernoulliPrior(size=shape, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ ch
             @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    armed0, used0 = ctx.prior_hint_stats()
    ep.iterate(max_iter=8, callback=T.PassCallback(), damping=0.1)
    armed1, used1 = ctx.prior_hint_stats()
    assert armed1 - armed0 == 8 and used1 - used0 == 7
    r = np.asarray(ep.get_variables_data()["x"]["r"]).reshape(B, n)
    fac = orc.ConvFactor(f)
    for b in range(B):
        res = orc.ep_chain([orc.GBPriorFactor(shape, rho=rho), fac, orc.GaussLikFactor(Y[b].reshape(shape), nv)],
                           [shape, shape], max_iter=8, damping=0.1, tol=0.0, max_increase=np.inf)
        np.testing.assert_allclose(r[b], res["r"][0].reshape(n), rtol=1e-6, atol=1e-8 * np.abs(res["r"][0]).max())


# ---- the likelihood twin (tramp_lik_hint): the channel's FORWARD update leaves sum v_i of the sign / abs likelihood
# that consumes its message (sgn_likelihood.py:32-37, abs_likelihood.py:31-36), the likelihood update is one pass
def _liks(T, rs, batch, m, dtype, ctx):
    z = rs.standard_normal((batch, m))
    return [("sgn", T.SgnLikelihood(y=np.sign(z), dtype=dtype, ctx=ctx), np.sign(z)),
            ("abs", T.AbsLikelihood(y=np.abs(z), dtype=dtype, ctx=ctx), np.abs(z))]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("batch", [1, 2, 5, 64, 70])
@pytest.mark.parametrize("damping", [0.0, 0.3])
def test_dense_producer_leaves_likelihood_moments(T, ctx, dtype, batch, damping):
    rs = np.random.RandomState(100 + batch)
    M, N = 9216, 40          # the message into z is longer than one workgroup's register tile
    W = rs.randn(M, N) / np.sqrt(N)
    ch = T.LinearChannel(W, dtype=dtype, ctx=ctx, svd="host")
    az = ctx.array(rs.uniform(0.5, 3, batch)); ax = ctx.array(rs.uniform(0.2, 5, batch))
    bz = ctx.array(rs.standard_normal((batch, N)), dtype); bx = ctx.array(rs.standard_normal((batch, M)), dtype)
    for name, lik, y in _liks(T, rs, batch, M, dtype, ctx):
        kind, yd = lik.lik_hint(batch, M)
        a_msg = ctx.array(rs.uniform(0.5, 2, batch))
        b_msg = ctx.array(rs.standard_normal((batch, M)), dtype)
        kw = dict(out_a=a_msg, out_b=b_msg)
        if damping:
            kw.update(damping=damping, old_a=a_msg, old_b=b_msg)
        armed0, used0 = ctx.prior_hint_stats()
        ctx._call("tramp_lik_hint", kind, M, batch, b_msg.ptr, a_msg.ptr, yd.ptr)
        ch.compute_forward_message(az, bz, ax, bx, **kw)
        a1, b1 = ctx.empty((batch,)), ctx.empty((batch, M), dtype)
        lik.compute_backward_message(a_msg, b_msg, out_a=a1, out_b=b1)     # fused: one pass over (bz, y)
        armed1, used1 = ctx.prior_hint_stats()
        assert (armed1 - armed0, used1 - used0) == (1, 1), (name, batch)
        a2, b2 = ctx.empty((batch,)), ctx.empty((batch, M), dtype)
        lik.compute_backward_message(a_msg, b_msg, out_a=a2, out_b=b2)     # hint consumed: the likelihood's own pass 1
        assert ctx.prior_hint_stats() == (armed1, used1)
        t = _tol(dtype)
        np.testing.assert_allclose(a1.get(), a2.get(), rtol=max(t["rtol"], 1e-12))
        np.testing.assert_allclose(b1.get(), b2.get(), rtol=t["rtol"], atol=t["atol"] * np.abs(b2.get()).max())
        if dtype == np.float64:
            am, bm = a_msg.get(), b_msg.get()
            for i in range(min(batch, 3)):
                a_ref, b_ref = (orc.sgn_lik_message if name == "sgn" else orc.abs_lik_message)(am[i], bm[i], y[i])
                np.testing.assert_allclose(a1.get()[i], a_ref, rtol=1e-9)
                np.testing.assert_allclose(b1.get()[i], b_ref, rtol=1e-8, atol=1e-9 * np.abs(b_ref).max())


def test_lik_hint_is_ignored_when_it_does_not_match(T, ctx):
    "another observation array or another likelihood: the likelihood runs its own first pass, same result"
    rs = np.random.RandomState(1)
    M, N, B = 4608, 24, 4
    ch = T.LinearChannel(rs.randn(M, N) / 5, ctx=ctx, svd="host")
    az = ctx.array(rs.uniform(1, 2, B)); ax = ctx.array(rs.uniform(1, 2, B))
    bz = ctx.array(rs.standard_normal((B, N))); bx = ctx.array(rs.standard_normal((B, M)))
    z = rs.standard_normal((B, M))
    sgn, sgn2, ab = (T.SgnLikelihood(y=np.sign(z), ctx=ctx), T.SgnLikelihood(y=-np.sign(z), ctx=ctx),
                     T.AbsLikelihood(y=np.abs(z), ctx=ctx))
    a_msg, b_msg = ctx.empty((B,)), ctx.empty((B, M))
    kind, yd = sgn.lik_hint(B, M)
    for other in (sgn2, ab):
        armed0, used0 = ctx.prior_hint_stats()
        ctx._call("tramp_lik_hint", kind, M, B, b_msg.ptr, a_msg.ptr, yd.ptr)
        ch.compute_forward_message(az, bz, ax, bx, out_a=a_msg, out_b=b_msg)
        a_f, b_f = other.compute_backward_message(a_msg.get(), b_msg.get())      # strict mode: no hint involved
        assert ctx.prior_hint_stats() == (armed0 + 1, used0)                       # (that call consumed it)
        ctx._call("tramp_lik_hint", kind, M, B, b_msg.ptr, a_msg.ptr, yd.ptr)
        ch.compute_forward_message(az, bz, ax, bx, out_a=a_msg, out_b=b_msg)
        a1, b1 = ctx.empty((B,)), ctx.empty((B, M))
        other.compute_backward_message(a_msg, b_msg, out_a=a1, out_b=b1)
        assert ctx.prior_hint_stats() == (armed0 + 2, used0)
        np.testing.assert_allclose(a1.get(), a_f, rtol=1e-13)
        np.testing.assert_allclose(b1.get(), b_f, rtol=1e-12, atol=1e-13)
    assert sgn.lik_hint(B, M + 1) is None and T.GaussianLikelihood(y=z, ctx=ctx).lik_hint(B, M) is None


_ONE_BIT = {}


def _one_bit_matrix(M, N):
    if (M, N) not in _ONE_BIT:
        W = orc.gaussian_ensemble(np.random.RandomState(7), M, N)
        _ONE_BIT[(M, N)] = (W, orc.LinearFactor(W, thin=True))
    return _ONE_BIT[(M, N)]


@pytest.mark.parametrize("batch", [1, 3, 64])
@pytest.mark.parametrize("output", ["sgn", "abs"])
def test_one_bit_ep_takes_the_fused_likelihood_path(T, ctx, batch, output):
    """1-bit CS / phase-retrieval-like EP (BASELINE configs[2]'s model): every iteration arms both ends of the chain
    (the backward update for the prior, the forward update for the likelihood); trajectories as the oracle's EP"""
    N, M, rho = 4352, 4608, 0.2                    # both messages past the single-workgroup tile
    W, lin = _one_bit_matrix(M, N)                 # one host SVD for the six parameter sets (it was 25 s of each 50 s run)
    rs = np.random.RandomState(7 + batch)
    X = np.stack([orc.gb_sample(rs, N, rho) for _ in range(batch)])
    Z = X @ W.T
    Y = np.sign(Z) if output == "sgn" else np.abs(Z)
    Lik = T.SgnLikelihood if output == "sgn" else T.AbsLikelihood
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.LinearChannel(W, ctx=ctx, svd="device") @ T.SISOVariable(id="z") @ Lik(y=Y, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    armed0, used0 = ctx.prior_hint_stats()
    n_it = 8
    ep.iterate(max_iter=n_it, callback=T.PassCallback(), damping=0.2)
    armed1, used1 = ctx.prior_hint_stats()
    assert armed1 - armed0 == 2 * n_it and used1 - used0 == 2 * n_it - 1, (armed1 - armed0, used1 - used0)
    data = ep.get_variables_data()
    LikF = orc.SgnLikFactor if output == "sgn" else orc.AbsLikFactor
    for b in range(min(batch, 2)):
        res = orc.ep_chain([orc.GBPriorFactor(N, rho=rho), lin, LikF(Y[b])], [N, M], max_iter=n_it, damping=0.2,
                           tol=0.0, max_increase=np.inf)
        for i, vid in enumerate(("x", "z")):
            r = np.atleast_2d(data[vid]["r"])[b]
            np.testing.assert_allclose(r, res["r"][i], rtol=1e-6, atol=1e-7 * np.abs(res["r"][i]).max())
            np.testing.assert_allclose(np.atleast_1d(data[vid]["v"])[b], res["v"][i], rtol=1e-7)


def test_conv_channel_inverse_fft_leaves_likelihood_moments(T, ctx):
    "the inverse FFT's fused message update as the producer of a sign likelihood's input (structured 1-bit CS)"
    rs = np.random.RandomState(6)
    n, B, rho = 16384, 3, 0.1
    ch = T.ConvChannel(rs.randn(n) / np.sqrt(n), ctx=ctx)
    X = rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n))
    Y = np.sign(ch.convolve(ctx.array(X)).get())
    model = (T.GaussBernoulliPrior(size=n, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ ch
             @ T.SISOVariable(id="z") @ T.SgnLikelihood(y=Y, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    armed0, used0 = ctx.prior_hint_stats()
    ep.iterate(max_iter=6, callback=T.PassCallback(), damping=0.2)
    armed1, used1 = ctx.prior_hint_stats()
    if armed1 == armed0:
        pytest.skip("this length runs on rocFFT (no fused message update)")
    assert armed1 - armed0 == 12 and used1 - used0 == 11
    r = ep.get_variables_data()["x"]["r"]
    fac = orc.ConvFactor(np.asarray(ch.filter))
    for b in range(B):
        res = orc.ep_chain([orc.GBPriorFactor(n, rho=rho), fac, orc.SgnLikFactor(Y[b])], [n, n],
                           max_iter=6, damping=0.2, tol=0.0, max_increase=np.inf)
        np.testing.assert_allclose(r[b], res["r"][0], rtol=1e-6, atol=1e-7 * np.abs(res["r"][0]).max())
