"""GPU parity of the unitary channels (DFTChannel, RotationChannel) against
vectors captured from the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def close(a, b, rtol, scale=None):
    b = np.asarray(b, dtype=float)
    sc = np.abs(b).max() if scale is None else scale
    np.testing.assert_allclose(np.asarray(a, dtype=float), b, rtol=rtol, atol=rtol * max(sc, 1e-300))


def make(T, ctx, c, dtype=np.float64):
    if c["kind"] == "dft":
        shape = c.get("shape")
        return T.DFTChannel(real=True, shape=shape if shape is not None and len(shape) > 1 else None,
                            dtype=dtype, ctx=ctx)
    return T.RotationChannel(c["R"], dtype=dtype, ctx=ctx)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_unitary_messages_match_reference(ctx, golden_unitary, dtype):
    import tramp_amd as T
    rt = 1e-12 if dtype == np.float64 else 1e-5
    for c in golden_unitary["msgs"]:
        ch = make(T, ctx, c, dtype)
        a, b = ch.compute_forward_message(1.3, c["bz"], 0.7, c["bx"])
        assert a == c["a_fwd"] and b.shape == c["b_fwd"].shape
        close(b, c["b_fwd"], rt)
        a, b = ch.compute_backward_message(1.3, c["bz"], 0.7, c["bx"])
        assert a == c["a_bwd"] and b.shape == c["b_bwd"].shape
        close(b, c["b_bwd"], rt)
        if c["kind"] == "dft":
            close(ch.sample(c["bz"]), c["x"], rt)
    # ensemble rows are independent
    c = golden_unitary["msgs"][0]
    ch = make(T, ctx, c)
    B = np.stack([c["bz"], 2 * c["bz"], -c["bz"]])
    a, b = ch.compute_forward_message(np.array([1.3, 2.0, 0.5]), B, 0.7, None)
    close(a, [1.3, 2.0, 0.5], 1e-15)
    close(b[1], 2 * c["b_fwd"], 1e-12)
    close(b[2], -c["b_fwd"], 1e-12)


def test_unitary_ep_trajectories(ctx, golden_unitary):
    import tramp_amd as T
    for c in golden_unitary["ep"]:
        N = c["N"]
        ch = make(T, ctx, c)
        model = (T.GaussBernoulliPrior(size=N, rho=c["rho"], ctx=ctx) @ T.SISOVariable(id="x") @ ch
                 @ T.SISOVariable(id="z")
                 @ T.GaussianLikelihood(y=c["y"].reshape(-1), var=c["noise_var"], ctx=ctx)).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx)
        recs = []

        def rec(algo, i, max_iter, stop=T.EarlyStoppingEP()):
            recs.append(algo.get_variables_data())
            return stop(algo, i, max_iter)
        ep.iterate(max_iter=30, callback=rec, damping=c["damping"])
        assert ep.n_iter == c["n_iter"], (c["kind"], c["damping"])
        t = c["traj"]
        for it in range(min(len(recs), t["v:x"].shape[0])):
            for vid in ("x", "z"):
                close(np.asarray(recs[it][vid]["r"]).reshape(-1), t[f"r:{vid}"][it].reshape(-1), 1e-7)
                close(recs[it][vid]["v"], t[f"v:{vid}"][it], 1e-7)


def test_dft_is_unitary_at_large_n(ctx):
    import tramp_amd as T
    n = 1 << 20
    rs = np.random.RandomState(0)
    z = rs.standard_normal(n)
    ch = T.DFTChannel(ctx=ctx)
    _, x = ch.compute_forward_message(1.0, z, 0.0, None)
    close(np.sum(x**2), np.sum(z**2), 1e-12)                    # Parseval
    _, zz = ch.compute_backward_message(0.0, None, 1.0, x)
    close(zz, z, 1e-12)
