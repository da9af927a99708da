"""GPU parity of the three-region likelihoods (HardTanh / HardSigmoid / SymmetricDoor; SURVEY 8f
rank 2) against vectors captured from tramp.likelihoods.piecewise_linear_likelihood."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu


def close(a, b, rtol, scale=None):
    b = np.asarray(b, dtype=float)
    sc = np.abs(b).max() if scale is None else scale
    np.testing.assert_allclose(np.asarray(a, dtype=float), b, rtol=rtol, atol=rtol * max(sc, 1e-300))


def make(T, ctx, kind, param, y, dtype=np.float64):
    kw = dict(dtype=dtype, ctx=ctx)
    if kind == "door":
        kw["width"] = param
    return T.get_likelihood(y, kind, **kw)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pwl3_lik_matches_reference(ctx, golden_pwl3_lik, dtype):
    import tramp_amd as T
    rt = 1e-9 if dtype == np.float64 else 2e-3
    for c in golden_pwl3_lik["cases"]:
        y = c["y"]
        if dtype == np.float32 and c["kind"] != "door":
            # a sloped region pins z = (y - x0)/slope: keep observations exactly representable in fp32
            y = y.astype(np.float32).astype(np.float64)
            ref_r, ref_v = orc.pwl3_lik_posterior(c["kind"], c["az"], c["bz"].astype(np.float32).astype(float), y,
                                                  param=c["param"])
        else:
            ref_r, ref_v = c["r"], c["v"]
        lk = make(T, ctx, c["kind"], c["param"], y, dtype)
        r, v = lk.compute_backward_posterior(c["az"], c["bz"], y)
        close(r, ref_r, rt)
        close(v, ref_v, rt * 10)
        if dtype == np.float64:
            a_new, b_new = lk.compute_backward_message(c["az"], c["bz"])
            # a_new = 1/v - az, b_new = r/v - bz cancel: compare on the scale of the terms
            close(a_new, c["a_new"], 1e-8, scale=c["az"] + 1 / max(c["v"], 1e-300))
            close(b_new, c["b_new"], 1e-8, scale=np.abs(c["r"]).max() / max(c["v"], 1e-300) + np.abs(c["bz"]).max())


def test_pwl3_lik_state_evolution(ctx, golden_pwl3_lik):
    import tramp_amd as T
    for c in golden_pwl3_lik["se"]:
        lk = make(T, ctx, c["kind"], c["param"], None)
        close(lk.compute_backward_error(c["az"], c["tau_z"]), c["v"], 1e-6)
    lk = T.HardTanhLikelihood(y=None, ctx=ctx)
    with pytest.raises(AssertionError):
        lk.compute_backward_error(0.5, 1.0)           # az <= 1/tau_z


def test_pwl3_lik_ep_and_to_observed(ctx, golden_pwl3_lik):
    """EP trajectories of perceptrons with door / hard-tanh outputs, built the reference's way:
    generative model with the CHANNEL, then to_observed swaps in the likelihood twin."""
    import tramp_amd as T
    for c in golden_pwl3_lik["ep"]:
        N, M = c["N"], c["M"]
        W = np.random.RandomState(c["seed"]).randn(M, N) / np.sqrt(N)
        assert W.sum() == c["W_sum"]
        out = T.SymmetricDoorChannel(width=c["param"], ctx=ctx) if c["kind"] == "door" else T.HardTanhChannel(ctx=ctx)
        gen = (T.GaussianPrior(size=N, ctx=ctx) @ T.SISOVariable(id="x") @ T.LinearChannel(W, ctx=ctx, svd="host")
               @ T.SISOVariable(id="z") @ out @ T.SILeafVariable(id="y")).to_model()
        model = gen.to_observed({"y": c["y"]})
        assert type(model.factors[-1]).__name__ == ("SymmetricDoorLikelihood" if c["kind"] == "door"
                                                    else "HardTanhLikelihood")
        ep = T.ExpectationPropagation(model, ctx=ctx)
        recs = []

        def rec(algo, i, max_iter, stop=T.EarlyStoppingEP()):
            recs.append(algo.get_variables_data())
            return stop(algo, i, max_iter)
        ep.iterate(max_iter=30, callback=rec, damping=c["damping"])
        assert ep.n_iter == c["n_iter"]
        t = c["traj"]
        for it in range(min(len(recs), t["v:x"].shape[0])):
            rt = 1e-6 if it < 6 else 1e-4
            for vid in ("x", "z"):
                close(recs[it][vid]["r"], t[f"r:{vid}"][it], rt)
                close(recs[it][vid]["v"], t[f"v:{vid}"][it], rt)


def test_pwl3_lik_incompatible_observation_is_nan(ctx):
    import tramp_amd as T
    y = np.array([1.0, 0.25, -1.0, 2.0])               # 2.0 is not a HardTanh output
    lk = T.HardTanhLikelihood(y=y, ctx=ctx)
    r, v = lk.compute_backward_posterior(1.3, np.array([0.3, -0.2, 0.5, 0.1]), y)
    assert np.isfinite(r[:3]).all() and np.isnan(r[3]) and r[1] == 0.25
