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
    with pytest.raises(NotImplementedError):
        T.HardTanhChannel(ctx=ctx).compute_log_partition(1.0, c["bz"], 1.0, c["bx"])


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
