"""GPU parity of the device-side EP bookkeeping (csrc/monitor.hip): per-instance stopping rule and NaN
guard of the reference's EarlyStoppingEP / check_message (tramp/algos/callbacks.py:250-286,
tramp/algos/message_passing.py:187-209) evaluated on the device behind the sweeps, the host polling
every k iterations.  Oracle: ep_chain on each instance alone (which, like the reference, goes back to
the previous iterate when it gives up after an increase)."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tramp_amd
    return tramp_amd


def _glm(T, ctx, B, N=256, M=128, rho=0.1, nv=1e-2, seed=31):
    rs = np.random.RandomState(seed)
    W = orc.gaussian_ensemble(rs, M, N)
    X = np.stack([orc.gb_sample(rs, N, rho) for _ in range(B)])
    Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((B, M))
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.LinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=Y if B > 1 else Y[0], var=nv, ctx=ctx)).to_model()
    fs = lambda b: [orc.GBPriorFactor(N, rho=rho), orc.LinearFactor(W), orc.GaussLikFactor(Y[b], nv)]   # noqa: E731
    return model, fs, X, (N, M)


def _close(a, b, rtol=1e-6):
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), b, rtol=rtol, atol=rtol * np.abs(b).max())


@pytest.mark.parametrize("batch", [1, 6])
def test_per_instance_rule_against_single_runs(T, ctx, batch):
    """tol and max_increase chosen so that the ensemble mixes instances that converge and instances that
    give up: each one must report what its own run would have (status, n_iter, r, v)."""
    model, fs, X, (N, M) = _glm(T, ctx, batch)
    rule = dict(tol=2e-3, wait_increase=5, max_increase=0.03)
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=100, callback=T.EarlyStoppingEP(**rule), damping=0.1)
    d = ep.get_variables_data()
    statuses = []
    for b in range(batch):
        res = orc.ep_chain(fs(b), [N, M], max_iter=100, damping=0.1, **rule)
        statuses.append(res["status"])
        assert {0: "max_iter", 1: "converged", 2: "diverged"}[int(ep.status[b])] == res["status"], b
        assert ep.n_iter_instance[b] == res["n_iter"], b
        r = np.atleast_2d(d["x"]["r"])[b]
        v = np.atleast_1d(d["x"]["v"])[b]
        _close(r, res["r"][0])
        _close(v, res["v"][0])
        _close(np.atleast_2d(d["z"]["r"])[b], res["r"][1])
    assert ep.n_iter == max(ep.n_iter_instance)
    if batch > 1:
        assert "diverged" in statuses       # the give-up branch (previous iterate restored) was exercised


@pytest.mark.parametrize("variant", ["check_every", "graph"])
def test_poll_cadence_and_graph_replay_do_not_change_results(T, ctx, variant):
    model, fs, X, (N, M) = _glm(T, ctx, 6)
    outs = []
    for kw in (dict(), dict(check_every=5) if variant == "check_every" else dict(graph=True)):
        ep = T.ExpectationPropagation(model, ctx=ctx)
        ep.iterate(max_iter=200, damping=0.1, **kw)
        d = ep.get_variables_data()
        outs.append((ep.n_iter, ep.status.copy(), ep.n_iter_instance.copy(), d["x"]["r"].copy(), d["x"]["v"].copy(),
                     d["z"]["r"].copy(), {k: v.copy() for k, v in ep.get_edges_data().items()}))
    a, b = outs
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.all(a[1] == 1) and len(set(a[2])) > 1            # all converged, at different iterations
    for i in (3, 4, 5):
        assert np.array_equal(a[i], b[i])
    for k in a[6]:
        assert np.array_equal(a[6][k], b[6][k])


def test_nan_guard_names_the_message(T, ctx):
    model, fs, X, (N, M) = _glm(T, ctx, 3)
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=2, callback=T.PassCallback())
    # poison one instance's message precision on the device, then go on
    A = ep.A.get()
    A[1, 2] = np.nan
    ep.A.set(A)
    with pytest.raises(ValueError, match=r"message into x \(instance 2\): a is nan"):
        ep.iterate(max_iter=3, callback=T.PassCallback(), warm_start=True, check_every=3)


def test_mse_validates_the_shape_of_the_true_signal(T, ctx):
    model, fs, X, (N, M) = _glm(T, ctx, 4)
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=5, callback=T.PassCallback())
    m = ep.mse("x", X)
    r = ep.get_variables_data()["x"]["r"]
    _close(m, np.mean((r - X)**2, axis=1), 1e-10)
    m1 = ep.mse("x", X[0])                                    # one signal shared by the ensemble
    _close(m1, np.mean((r - X[0])**2, axis=1), 1e-10)
    for bad in (X[:2], X[:, :N - 1], np.zeros(N + 1)):
        with pytest.raises(ValueError):
            ep.mse("x", bad)


@pytest.mark.parametrize("batch", [1, 6])
def test_stopped_instances_get_their_messages_back(T, ctx, batch):
    """"restoring old message dag" (tramp/algos/callbacks.py:277-283) per instance: after the run every instance
    holds the MESSAGES its own run would have left - those of its stopping iteration if it converged, those of the
    iteration before the increase if it gave up - although the ensemble kept iterating for the others."""
    model, fs, X, (N, M) = _glm(T, ctx, batch)
    rule = dict(tol=2e-3, wait_increase=5, max_increase=0.03)
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=100, callback=T.EarlyStoppingEP(**rule), damping=0.1)
    assert ep._restore
    seen = set()
    edges = ep.get_edges_data()
    for b in range(batch):
        res = orc.ep_chain(fs(b), [N, M], max_iter=100, damping=0.1, **rule)
        seen.add(res["status"])
        for i, vid in enumerate(("x", "z")):
            for direction, ref in (("fwd", res["fwd"][i]), ("bwd", res["bwd"][i])):
                a, bmsg = ep.get_message(vid, direction)
                _close(np.atleast_1d(a)[b], ref[0], 1e-7)
                _close(np.atleast_2d(bmsg)[b], ref[1], 1e-6)
                _close(np.atleast_1d(edges[(vid, direction)])[b], ref[0], 1e-7)
    if batch > 1:
        assert seen == {"converged", "diverged"}
    # and a warm start goes on from there: the same as a fresh run of the oracle continued from its final state is
    # not available, but the restored state must at least be a fixed input: two warm starts of one iteration agree
    m0 = [ep.get_message(v, d)[1].copy() for v in ("x", "z") for d in ("fwd", "bwd")]
    ep2 = T.ExpectationPropagation(model, ctx=ctx)
    ep2.iterate(max_iter=100, callback=T.EarlyStoppingEP(**rule), damping=0.1, restore_messages=False)
    m1 = [ep2.get_message(v, d)[1] for v in ("x", "z") for d in ("fwd", "bwd")]
    if batch > 1:
        assert any(np.abs(a - b).max() > 1e-9 for a, b in zip(m0, m1))      # without the restore the live messages moved on


def test_nan_restores_the_previous_messages(T, ctx):
    "a NaN found after an iteration: every instance goes back to the messages before it, then ValueError (message_passing.py:187-209)"
    model, fs, X, (N, M) = _glm(T, ctx, 3)
    ref = T.ExpectationPropagation(model, ctx=ctx)
    ref.iterate(max_iter=2, callback=T.PassCallback(), damping=0.1)
    want = {(v, d): ref.get_message(v, d) for v in ("x", "z") for d in ("fwd", "bwd")}
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=2, callback=T.PassCallback(), damping=0.1, restore_messages=True)
    lik = model.next_factor(ep.variables[0])                # the channel: its backward update runs every iteration
    real = lik.compute_backward_message
    calls = []

    def poisoned(*args, **kw):
        out = real(*args, **kw)
        calls.append(1)
        if "out_a" in kw:
            a = kw["out_a"].get()
            a[1] = np.nan
            kw["out_a"].set(a)
        return out
    lik.compute_backward_message = poisoned
    try:
        with pytest.raises(ValueError, match="a is nan"):
            ep.iterate(max_iter=1, callback=T.PassCallback(), damping=0.1, warm_start=True, restore_messages=True)
    finally:
        del lik.compute_backward_message
    assert calls
    for (v, d), (a_ref, b_ref) in want.items():
        a, b = ep.get_message(v, d)
        np.testing.assert_array_equal(a, a_ref)
        np.testing.assert_array_equal(b, b_ref)


def test_device_array_dropped_during_capture_does_not_invalidate_it(T, ctx):
    """Python may garbage-collect a device array at any moment, also between tramp_graph_begin and tramp_graph_end:
    its free (a stream synchronisation + hipFree) waits for the end of the capture instead of invalidating it."""
    import ctypes as C
    a, b = ctx.zeros((4, 1000)), ctx.zeros((4, 1000))
    victim = ctx.zeros((8, 4096))
    channel = T.LinearChannel(np.random.RandomState(0).randn(6, 9), ctx=ctx, svd="host")
    event = ctx.event()
    b.copy_from(a)
    ctx.sync()
    ctx._call("tramp_graph_begin")
    try:
        a.fill(2.0)
        del victim                                   # -> DeviceArray.__del__ -> tramp_buf_free inside the capture
        del channel, event                           # -> tramp_linear_destroy / tramp_event_destroy: held until the end
        b.copy_from(a)
    finally:
        g = C.c_void_p()
        ctx._call("tramp_graph_end", C.byref(g))
    ctx._call("tramp_graph_launch", g)
    ctx.sync()
    ctx._call("tramp_graph_destroy", g)
    assert np.all(b.get() == 2.0)


@pytest.mark.parametrize("damping", [None, 0.3])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("kind", ["chain", "conv", "tree"])
def test_alternating_message_buffers_change_no_bit(T, ctx, kind, graph, damping):
    """restore_messages keeps the previous iteration's messages by alternating two sets of buffers (algos.py
    _take_snapshot, tree.py _swap_sets) instead of copying: with and without it every message and posterior has the same
    bits after an odd and an even number of iterations, eager and as a hipGraph replay, with and without damping
    (Gaussian factors: the constant-message and projection-cache paths)."""
    rs = np.random.RandomState(17)
    if kind == "chain":
        model, _, _, _ = _glm(T, ctx, 3)
        ids = ["x", "z"]
    elif kind == "conv":
        n, B = 4096, 2
        f = rs.standard_normal(n) / 64
        Y = rs.standard_normal((B, n))
        model = (T.GaussianPrior(size=n, ctx=ctx) @ T.SISOVariable(id="x") @ T.ConvChannel(f, ctx=ctx)
                 @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=Y, var=0.05, ctx=ctx)).to_model()
        ids = ["x", "z"]
    else:
        from tests.test_gpu_tree import build_tree
        model = build_tree(T, ctx, dict(shape=(64,), noise_var=0.01, rho=0.1), y=rs.standard_normal((2, 64)))
        ids = ["x", "z"]
    for n_it in (6, 7):
        out = {}
        for restore in (False, True):
            ep = T.ExpectationPropagation(model, ctx=ctx)
            ep.iterate(max_iter=n_it, callback=T.EarlyStoppingEP(tol=1e-300), damping=damping, graph=graph,
                       restore_messages=restore)
            d = ep.get_variables_data()
            rec = {("r", v): np.asarray(d[v]["r"]).copy() for v in ids}
            rec.update({("v", v): np.asarray(d[v]["v"]).copy() for v in ids})
            rec.update({("a",) + k: np.asarray(a).copy() for k, a in ep.get_edges_data().items()})
            if kind != "tree":
                for v in ids:
                    for direction in ("fwd", "bwd"):
                        rec[("b", v, direction)] = np.asarray(ep.get_message(v, direction)[1]).copy()
            out[restore] = rec
        assert out[False].keys() == out[True].keys()
        for k in out[False]:
            assert np.array_equal(out[False][k], out[True][k]), (kind, graph, damping, n_it, k)
