"""EP with damping="adaptive" (tramp/algos/message_passing.py:129-185) on the device scheduler against trajectories
captured from the reference (tests/golden/adaptive.npz, tools/make_golden_adaptive.py): all four messages of every
variable - factor -> variable and the separately damped variable -> factor edges - and the posteriors, iteration by
iteration.  The goldens keep only configurations whose step decisions do not hinge on round-off."""
import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def close(a, b, rtol, scale=None):
    b = np.asarray(b, dtype=float)
    sc = np.abs(b).max() if scale is None else scale
    np.testing.assert_allclose(np.asarray(a, dtype=float), b, rtol=rtol, atol=rtol * max(sc, 1e-300))


def build(T, ctx, c, B=1):
    N, M = c["N"], int(c["alpha"] * c["N"])
    rs = np.random.RandomState(c["seed"])
    W = rs.randn(M, N) / np.sqrt(N)
    assert W.sum() == c["W_sum"]
    y = np.tile(c["y"], (B, 1)) if B > 1 else c["y"]
    if c["output"] == "gaussian":
        lik = T.GaussianLikelihood(y=y, var=0.01, ctx=ctx)
    elif c["output"] == "sgn":
        lik = T.SgnLikelihood(y=y, ctx=ctx)
    else:
        lik = T.AbsLikelihood(y=y, ctx=ctx)
    return (T.GaussBernoulliPrior(size=N, rho=c["rho"], ctx=ctx) @ T.SISOVariable(id="x")
            @ T.LinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z") @ lik).to_model()


def edges_of(ep):
    out = {}
    for i, v in enumerate(ep.variables):
        out[f"fwd:f->{v.id}"] = (ep.a_fwd[i].get(), ep.b_fwd[i].get())
        out[f"bwd:f->{v.id}"] = (ep.a_bwd[i].get(), ep.b_bwd[i].get())
        out[f"fwd:{v.id}->f"] = (ep.a_vf[i].get(), ep.b_vf[i].get())
        out[f"bwd:{v.id}->f"] = (ep.a_vb[i].get(), ep.b_vb[i].get())
    return out


@pytest.mark.parametrize("B", [1, 3])
def test_adaptive_damping_trajectories(ctx, B):
    import tramp_amd as T
    for c in load_golden("adaptive"):
        model = build(T, ctx, c, B)
        ep = T.ExpectationPropagation(model, ctx=ctx)
        recs = []

        def rec(algo, i, max_iter):
            recs.append((edges_of(algo), algo.get_variables_data()))
            return False

        ep.iterate(max_iter=c["n_iter"], callback=rec, damping="adaptive")
        assert len(recs) == c["n_iter"]
        for it, (edges, data) in enumerate(recs):
            t = c["traj"][it]
            rt = 1e-8 if it < 3 else 1e-6
            for vid in ("x", "z"):
                a_tot = t[f"fwd:f->{vid}"][0] + t[f"bwd:f->{vid}"][0]
                for key in (f"fwd:f->{vid}", f"bwd:f->{vid}", f"fwd:{vid}->f", f"bwd:{vid}->f"):
                    ra, rb = t[key]
                    a, b = edges[key]
                    for inst in range(B):
                        close(a[inst], ra, rt, scale=max(a_tot, abs(ra)))
                        close(b[inst], rb, rt, scale=max(np.abs(rb).max(), np.abs(t[f"r:{vid}"]).max() * a_tot))
                r = np.asarray(data[vid]["r"]).reshape(B, -1)
                v = np.atleast_1d(data[vid]["v"])
                for inst in range(B):
                    close(r[inst], t[f"r:{vid}"], rt)
                    close(v[inst], t[f"v:{vid}"], rt)


def test_adaptive_damping_refuses_graph_and_keeps_float_damping(ctx):
    import tramp_amd as T
    c = load_golden("adaptive")[0]
    ep = T.ExpectationPropagation(build(T, ctx, c), ctx=ctx)
    with pytest.raises(ValueError, match="hipGraph"):
        ep.iterate(max_iter=2, damping="adaptive", graph=True)
    with pytest.raises(ValueError, match="damping must be"):
        ep.iterate(max_iter=2, damping="strong")
    ep.iterate(max_iter=3, damping=0.2)                       # the constant-damping path is untouched
    assert ep.n_iter == 3
