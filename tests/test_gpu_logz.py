"""GPU parity of the log-partitions and the EP log-evidence against values
captured from the reference's compute_log_partition / log_evidence."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu


def close(a, b, rtol, atol=0.0):
    np.testing.assert_allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_factor_log_partitions_match_reference(ctx, golden_logz, dtype):
    import tramp_amd as T
    rt = 1e-11 if dtype == np.float64 else 2e-4
    kw = dict(dtype=dtype, ctx=ctx)
    for c in golden_logz["cases"]:
        az, ax, bz, bx = c["az"], c["ax"], c["bz"], c["bx"]
        n = bz.size
        sc = 1.0 if dtype == np.float64 else max(1.0, az, ax)          # fp32 inputs: errors scale with a b^2 / a
        close(T.GaussBernoulliPrior(size=n, rho=0.1, mean=0.3, var=1.7, **kw).compute_log_partition(ax, bx),
              c["gb"], rt * sc, rt)
        close(T.GaussianPrior(size=n, mean=0.3, var=1.7, **kw).compute_log_partition(ax, bx), c["gauss_prior"],
              rt * sc, rt)
        close(T.MAP_L1NormPrior(size=n, gamma=0.8, **kw).compute_log_partition(ax, bx), c["l1"], rt * sc, rt)
        close(T.GaussianLikelihood(y=c["y_g"], var=0.4, **kw).compute_log_partition(az, bz, c["y_g"]),
              c["gauss_lik"], rt * sc, rt)
        close(T.SgnLikelihood(y=c["y_sgn"], **kw).compute_log_partition(az, bz, c["y_sgn"]), c["sgn_lik"], rt * sc, rt)
        close(T.AbsLikelihood(y=c["y_abs"], **kw).compute_log_partition(az, bz, c["y_abs"]), c["abs_lik"], rt * sc, rt)
        for name, cls in (("sgn", T.SgnChannel), ("abs", T.AbsChannel), ("relu", T.ReluChannel)):
            close(cls(**kw).compute_log_partition(az, bz, ax, bx), c["pwl_" + name], rt * sc * 10, rt * n)
    # ensemble rows are independent instances
    c = golden_logz["cases"][0]
    p = T.GaussBernoulliPrior(size=c["bx"].size, rho=0.1, mean=0.3, var=1.7, ctx=ctx)
    A = p.compute_log_partition(np.array([c["ax"], 2 * c["ax"]]), np.stack([c["bx"], c["bx"]]))
    close(A[0], c["gb"], 1e-11)
    close(A[1], orc.gb_prior_log_partition(2 * c["ax"], c["bx"], 0.1, 0.3, 1.7), 1e-11)


def test_linear_log_partition_matches_reference(ctx, golden_logz):
    import tramp_amd as T
    for c in golden_logz["lin"]:
        ch = T.LinearChannel(c["W"], ctx=ctx, svd="host")
        for k in c["cases"]:
            close(ch.compute_log_partition(k["az"], c["bz"], k["ax"], c["bx"]), k["logZ"], 1e-9, 1e-9)


def test_ep_log_evidence_matches_reference(ctx, golden_logz):
    import tramp_amd as T
    for c in golden_logz["ep"]:
        if c["kind"] == "glm":
            N, M = c["N"], int(c["alpha"] * c["N"])
            W = np.random.RandomState(c["seed"]).randn(M, N) / np.sqrt(N)
            lik = {"gaussian": lambda: T.GaussianLikelihood(y=c["y"], var=c["noise_var"], ctx=ctx),
                   "sgn": lambda: T.SgnLikelihood(y=c["y"], ctx=ctx),
                   "abs": lambda: T.AbsLikelihood(y=c["y"], ctx=ctx)}[c["output"]]()
            model = (T.GaussBernoulliPrior(size=N, rho=c["rho"], ctx=ctx) @ T.SISOVariable(id="x")
                     @ T.LinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z") @ lik).to_model()
        else:
            model = T.MultiLayerModel([T.GaussianPrior(size=c["N"], ctx=ctx), T.LinearChannel(c["W1"], ctx=ctx, svd="host"),
                                       T.ReluChannel(ctx=ctx), T.LinearChannel(c["W2"], ctx=ctx, svd="host"),
                                       T.GaussianLikelihood(y=c["y"], var=c["noise_var"], ctx=ctx)],
                                      ids=["x", "z1", "a1", "z2", "y"])
        ep = T.ExpectationPropagation(model, ctx=ctx)
        As = []

        def cb(algo, i, max_iter):
            As.append(algo.log_evidence())
            return False
        ep.iterate(max_iter=len(c["A"]), callback=cb, damping=c["damping"])
        close(As, c["A"], 1e-6, 1e-7 * np.abs(c["A"]).max())
        close(ep.surprisal(), -c["A"][-1], 1e-6, 1e-7 * np.abs(c["A"]).max())


F2_CHANNELS = {"l-relu": lambda T, kw, p: T.LeakyReluChannel(p["slope"], **kw),
               "a-abs": lambda T, kw, p: T.AsymmetricAbsChannel(p["shift"], **kw),
               "h-tanh": lambda T, kw, p: T.HardTanhChannel(**kw),
               "h-sigm": lambda T, kw, p: T.HardSigmoidChannel(**kw),
               "door": lambda T, kw, p: T.SymmetricDoorChannel(p["width"], **kw)}
F2_LIKS = {"relu": lambda T, y, kw, p: T.ReluLikelihood(y=y, **kw),
           "l-relu": lambda T, y, kw, p: T.LeakyReluLikelihood(p["slope"], y=y, **kw),
           "h-tanh": lambda T, y, kw, p: T.HardTanhLikelihood(y=y, **kw),
           "h-sigm": lambda T, y, kw, p: T.HardSigmoidLikelihood(y=y, **kw),
           "door": lambda T, y, kw, p: T.SymmetricDoorLikelihood(p["width"], y=y, **kw),
           "a-abs": lambda T, y, kw, p: T.AsymmetricAbsLikelihood(y=y, shift=p["shift"], **kw)}


def test_f2_family_log_partitions_match_reference(ctx, golden_f2_logz):
    """LeakyRelu / AsymmetricAbs / HardTanh / HardSigmoid / SymmetricDoor channels and the observed twins against
    the reference's compute_log_partition (piecewise_linear_channel.py:55-58, piecewise_linear_likelihood.py:216-219)"""
    import tramp_amd as T
    kw = dict(ctx=ctx)
    for c in golden_f2_logz:
        if c["family"] == "channel":
            ch = F2_CHANNELS[c["name"]](T, kw, c["kwargs"])
            close(ch.compute_log_partition(c["az"], c["bz"], c["ax"], c["bx"]), c["A"], 1e-10, 1e-9)
        elif c["name"] != "a-abs":
            lk = F2_LIKS[c["name"]](T, c["y"], kw, c["kwargs"])
            close(lk.compute_log_partition(c["az"], c["bz"], c["y"]), c["A"], 1e-10, 1e-11)
    # f32 inputs, and the ensemble axis
    c = next(k for k in golden_f2_logz if k["family"] == "channel" and k["name"] == "h-tanh")
    ch32 = T.HardTanhChannel(dtype=np.float32, ctx=ctx)
    close(ch32.compute_log_partition(c["az"], c["bz"], c["ax"], c["bx"]), c["A"], 2e-4)
    A2 = T.HardTanhChannel(ctx=ctx).compute_log_partition(np.array([c["az"], 2.0]), np.stack([c["bz"], c["bz"]]),
                                                          np.array([c["ax"], 0.3]), np.stack([c["bx"], c["bx"]]))
    close(A2[0], c["A"], 1e-10)
    close(A2[1], orc.region_channel_log_partition(orc.f2_regions("h-tanh"), 2.0, c["bz"], 0.3, c["bx"]), 1e-10)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_asymmetric_abs_likelihood(ctx, golden_f2_logz, dtype):
    """AsymmetricAbsLikelihood (piecewise_linear_likelihood.py:260-265, registry key "a-abs").  The reference's own
    outputs are NaN for every input (log of the descending branch's slope, :59) - pinned in the golden and reproduced
    by reference_nan=True; the kernel takes the Jacobian |slope| and is checked against the oracle's restatement of
    the same region loop with that one change."""
    import tramp_amd as T
    rt = dict(rtol=1e-9, atol=1e-10) if dtype == np.float64 else dict(rtol=1e-3, atol=1e-3)
    for c in (k for k in golden_f2_logz if k["family"] == "likelihood" and k["name"] == "a-abs"):
        shift, az, bz, y = c["kwargs"]["shift"], c["az"], c["bz"], c["y"]
        strict = T.AsymmetricAbsLikelihood(y=y, shift=shift, reference_nan=True, dtype=dtype, ctx=ctx)
        r, v = strict.compute_backward_posterior(az, bz, y)
        assert np.isnan(r).all() and np.isnan(v) and np.isnan(c["r"]).all() and np.isnan(c["v"])
        assert np.isnan(strict.compute_log_partition(az, bz, y)) and np.isnan(c["A"])
        lk = T.get_likelihood(y, "a-abs", shift=shift, dtype=dtype, ctx=ctx)
        assert isinstance(lk, T.AsymmetricAbsLikelihood)
        reg = orc.f2_regions("a-abs", shift=float(dtype(shift)))          # the kink as the working precision holds it
        r, v = lk.compute_backward_posterior(az, bz, y)
        r_ref, v_ref = orc.region_lik_posterior(reg, az, bz.astype(dtype).astype(float), y.astype(dtype).astype(float),
                                                abs_slope=True)
        nan = np.isnan(r_ref)
        assert nan.sum() >= 1 and np.array_equal(np.isnan(r), nan)      # y <= -shift: no compatible branch
        np.testing.assert_allclose(r[~nan], r_ref[~nan], **rt)
        assert np.isnan(v) and np.isnan(v_ref)                          # the mean over an instance with such a y
        ok = ~nan
        lk2 = T.AsymmetricAbsLikelihood(y=y[ok], shift=shift, dtype=dtype, ctx=ctx)
        a_new, b_new = lk2.compute_backward_message(az, bz[ok])
        a_ref, b_ref = orc.aabs_lik_message(az, bz[ok].astype(dtype).astype(float), y[ok].astype(dtype).astype(float),
                                            shift=float(dtype(shift)))
        np.testing.assert_allclose(a_new, a_ref, rtol=rt["rtol"] * 10)
        np.testing.assert_allclose(b_new, b_ref, rtol=rt["rtol"], atol=rt["atol"] * 10 * np.abs(b_ref).max())
        A = lk2.compute_log_partition(az, bz[ok], y[ok])
        np.testing.assert_allclose(A, orc.region_lik_log_partition(reg, az, bz[ok].astype(dtype).astype(float),
                                                                  y[ok].astype(dtype).astype(float), abs_slope=True),
                                   rtol=1e-9 if dtype == np.float64 else 2e-4)
    # diagonal beliefs (the mode the reference's factor tests run in) and the SE error
    rs = np.random.RandomState(8)
    n = 64
    y = np.abs(rs.standard_normal(n)) + 0.4
    a = np.linspace(1, 2, n); b = np.linspace(-2, 2, n)
    lk = T.AsymmetricAbsLikelihood(y=y, shift=0.3, isotropic=False, dtype=dtype, ctx=ctx)
    r, v = lk.compute_backward_posterior(a, b, y)
    r_ref, v_ref = orc.abs_lik_posterior(a, b.astype(dtype).astype(float), y.astype(dtype).astype(float), isotropic=False)
    np.testing.assert_allclose(r, r_ref, **rt); np.testing.assert_allclose(v, v_ref, **rt)
    if dtype == np.float64:
        for shift in (1e-4, 0.3):
            lk = T.AsymmetricAbsLikelihood(y=None, shift=shift, ctx=ctx)
            np.testing.assert_allclose(lk.compute_backward_error(1.7, 1.0), orc.se_aabs_lik_error(1.7, 1.0, shift=shift),
                                       rtol=2e-6)
        with pytest.raises(AssertionError):
            T.AsymmetricAbsLikelihood(y=None, ctx=ctx).compute_backward_error(0.5, 1.0)


def test_observed_asymmetric_abs_channel_becomes_the_likelihood(ctx):
    "Model.to_observed / channel2likelihood (tramp/models/dag_algebra.py:26-27)"
    import tramp_amd as T
    from tramp_amd.models import channel2likelihood
    y = np.abs(np.random.RandomState(0).standard_normal(16))
    lk = channel2likelihood(T.AsymmetricAbsChannel(shift=0.05, ctx=ctx), y=y, y_name="y")
    assert isinstance(lk, T.AsymmetricAbsLikelihood) and lk.shift == 0.05
