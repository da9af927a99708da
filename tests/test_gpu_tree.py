"""GPU parity of the gradient / unitary channels and of EP on tree-structured models (a variable with
two children, a leaf with two parents) against vectors captured from the reference
(tools/make_golden.py `tree`: GradientChannel, UnitaryChannel, the sparse-gradient model of
examples/sparse/plot_sparse_gradient.py / models/total_variation_model.py) and the CPU oracle."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu

TREE_EDGES = [("GaussianPrior", "x"), ("x", "GaussianLikelihood"), ("x", "GradientChannel"),
              ("GradientChannel", "z"), ("GaussBernoulliPrior", "z")]


def close(a, b, rtol, atol=0.0):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.shape != b.shape and a.size == b.size:
        # estimates come in the variable's own shape ((d,) + shape gradient fields, images: test_estimates_keep_the_variable_shape),
        # the oracle's and some goldens are flat: values are compared in memory order
        a, b = a.ravel(), b.ravel()
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_estimates_keep_the_variable_shape(ctx):
    "get_variables_data hands r out as the reference does: in the variable's shape, an ensemble with its instance axis in front"
    import tramp_amd as T
    from tramp_amd.variables import SIMOVariable, MILeafVariable
    rs = np.random.RandomState(0)
    shape = (6, 8)
    for B in (1, 3):
        y = rs.standard_normal((B,) + shape) if B > 1 else rs.standard_normal(shape)
        model = (T.GaussianPrior(size=shape, ctx=ctx) @ SIMOVariable(id="x", n_next=2) @ (
            T.GaussianLikelihood(y=y, var=0.1, ctx=ctx) + (
                T.GradientChannel(shape=shape, ctx=ctx) + T.GaussBernoulliPrior(size=(2,) + shape, rho=0.3, ctx=ctx)
            ) @ MILeafVariable(id="z", n_prev=2))).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx).iterate(max_iter=5, damping=0.2)
        assert ep.batch == B
        d = ep.get_variables_data()
        lead = () if B == 1 else (B,)
        assert d["x"]["r"].shape == lead + shape and d["z"]["r"].shape == lead + (2,) + shape
        assert np.shape(d["x"]["v"]) == lead


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gradient_channel_matches_reference(ctx, golden_tree, dtype):
    import tramp_amd as T
    rt, at = (1e-10, 1e-10) if dtype == np.float64 else (2e-4, 2e-4)
    for g in golden_tree["grads"]:
        shape = tuple(g["shape"])
        ch = T.GradientChannel(shape=shape, dtype=dtype, ctx=ctx)
        close(ch.second_moment(1.7), g["second_moment"], 1e-12)
        close(ch.sample(g["z"]), g["x"], rt, at)
        sc = float(np.abs(g["bx"]).max())
        for c in g["cases"]:
            az, ax = c["az"], c["ax"]
            cond = max(1.0, az, ax)                     # messages are r (a + a_new) - b
            close(ch.compute_n_eff(az, ax), c["n_eff"], 1e-12, 1e-15)
            rz, vz = ch.compute_backward_posterior(az, g["bz"], ax, g["bx"])
            rx, vx = ch.compute_forward_posterior(az, g["bz"], ax, g["bx"])
            # vz = (1 - n_eff) / az loses eps / az when n_eff -> 1 (see the conv tests)
            close(vz, c["vz"], max(1e-9, rt), 4e-16 / az)
            close(vx, c["vx"], max(1e-12, rt))
            scale_z = max(1.0, float(np.abs(c["rz"]).max()))
            close(rz, c["rz"], rt, at * scale_z)
            close(rx, c["rx"], rt, at * max(1.0, float(np.abs(c["rx"]).max())))
            assert rx.shape == (len(shape),) + shape and rz.shape == shape
            if dtype == np.float64 and 1e-6 < az < 1e6 and 1e-6 < ax < 1e6:
                fa, fb = ch.compute_forward_message(az, g["bz"], ax, g["bx"])
                ba, bb = ch.compute_backward_message(az, g["bz"], ax, g["bx"])
                close(fa, c["a_fwd"], 1e-9)
                close(ba, c["a_bwd"], 1e-9)
                close(fb, c["b_fwd"], 1e-8, 1e-8 * sc * cond)
                close(bb, c["b_bwd"], 1e-8, 1e-8 * sc * cond)


def test_gradient_channel_ensemble_and_damping(ctx):
    "rows of an ensemble are independent instances; fused damping equals damping the plain message"
    import tramp_amd as T
    rs = np.random.RandomState(3)
    shape, B = (12, 10), 3
    ch = T.GradientChannel(shape=shape, ctx=ctx)
    o = orc.GradientOracle(shape)
    az, ax = rs.uniform(0.5, 2, B), rs.uniform(0.5, 2, B)
    bz, bx = rs.standard_normal((B,) + shape), rs.standard_normal((B, 2) + shape)
    fa, fb = ch.compute_forward_message(az, bz, ax, bx)
    ba, bb = ch.compute_backward_message(az, bz, ax, bx)
    for b in range(B):
        a, m = o.forward_message(az[b], bz[b], ax[b], bx[b])
        close(fa[b], a, 1e-10)
        close(fb[b], m, 1e-9, 1e-10)
        a, m = o.backward_message(az[b], bz[b], ax[b], bx[b])
        close(ba[b], a, 1e-10)
        close(bb[b], m, 1e-9, 1e-10)
    d = 0.3
    old_a = ctx.array(rs.uniform(1, 2, B), np.float64)
    old_b = ctx.array(rs.standard_normal((B, 2 * 120)), np.float64)
    want_a, want_b = d * old_a.get() + (1 - d) * fa, d * old_b.get().reshape((B, 2) + shape) + (1 - d) * fb
    ch.compute_forward_message(ctx.array(az, np.float64), ctx.array(bz.reshape(B, -1), np.float64),
                               ctx.array(ax, np.float64), ctx.array(bx.reshape(B, -1), np.float64),
                               out_a=old_a, out_b=old_b, damping=d, old_a=old_a, old_b=old_b)
    close(old_a.get(), want_a, 1e-12)
    close(old_b.get().reshape((B, 2) + shape), want_b, 1e-11, 1e-12)


def test_gradient_channel_normal_equations_large(ctx):
    "n = 2^20 (1-D) and 1024 x 1024: (az + ax s) rz_fft = fz + sum_i conj(w_i)^* ... checked on the device result"
    import tramp_amd as T
    rs = np.random.RandomState(5)
    for shape in [(1 << 20,), (1024, 1024)]:
        ch = T.GradientChannel(shape=shape, ctx=ctx)
        o = orc.GradientOracle(shape)
        d = len(shape)
        az, ax = 1.3, 0.7
        bz, bx = rs.standard_normal(shape), rs.standard_normal((d,) + shape)
        rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
        rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
        axes = tuple(range(1, d + 1))
        lhs = (az + ax * o.spectrum) * np.fft.fftn(rz)
        rhs = np.fft.fftn(bz) + (o.w_fft_bar * np.fft.fftn(bx, axes=axes)).sum(axis=0)
        assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(rhs).max()
        close(rx, o.convolve(rz), 1e-9, 1e-10)                         # rx = grad rz
        close(az * vz + ax * d * vx, 1.0, 1e-10)                       # az vz + ax d vx = (1 - n_eff) + n_eff


def test_unitary_channel_matches_reference(ctx, golden_tree):
    import tramp_amd as T
    for u in golden_tree["unit"]:
        ch = T.UnitaryChannel(u["U"], ctx=ctx)
        a, b = ch.compute_forward_message(1.3, u["bz"], 0.7, u["bx"])
        assert a == u["a_fwd"] and b.shape == u["b_fwd"].shape
        close(b, u["b_fwd"], 1e-12, 1e-13)
        a, b = ch.compute_backward_message(1.3, u["bz"], 0.7, u["bx"])
        assert a == u["a_bwd"]
        close(b, u["b_bwd"], 1e-12, 1e-13)
        close(ch.sample(u["bz"]), u["x"], 1e-12, 1e-13)
    with pytest.raises(ValueError):
        T.UnitaryChannel(np.ones((3, 3)), ctx=ctx)


def build_tree(T, ctx, e, y=None, dtype=np.float64):
    from tramp_amd.variables import SIMOVariable, MILeafVariable
    shape = tuple(e["shape"])
    d = len(shape)
    kw = dict(dtype=dtype, ctx=ctx)
    y = e["y"] if y is None else y
    n = int(np.prod(shape))
    return (T.GaussianPrior(size=shape, **kw) @ SIMOVariable(id="x", n_next=2) @ (
        T.GaussianLikelihood(y=np.asarray(y).reshape(-1, n) if np.ndim(y) > d else np.asarray(y).reshape(n),
                             var=e["noise_var"], **kw) + (
            T.GradientChannel(shape=shape, **kw) + T.GaussBernoulliPrior(size=(d,) + shape, rho=e["rho"], **kw)
        ) @ MILeafVariable(id="z", n_prev=2))).to_model()


class TreeRecorder:
    def __init__(self):
        self.iters = []

    def __call__(self, algo, i, max_iter):
        from tramp_amd.base import Variable
        rec = {}
        for (src, dst), m in algo.msg.items():
            key = f"{src.id if isinstance(src, Variable) else type(src).__name__}->" \
                  f"{dst.id if isinstance(dst, Variable) else type(dst).__name__}"
            rec[key + ":a"] = m.a.get().copy()
            rec[key + ":b"] = m.b.get().copy()
        for vid, d in algo.get_variables_data().items():
            rec["r:" + vid], rec["v:" + vid] = np.array(d["r"]), np.array(d["v"])
        self.iters.append(rec)
        return False


def test_tree_ep_trajectories_match_reference(ctx, golden_tree):
    """Sparse-gradient model: every message of both kinds and every posterior after each of the first 12
    iterations, against the reference's own run."""
    import tramp_amd as T
    from tramp_amd.tree import TreeExpectationPropagation
    for e in golden_tree["ep"]:
        model = build_tree(T, ctx, e)
        ep = T.ExpectationPropagation(model, ctx=ctx)
        assert isinstance(ep, TreeExpectationPropagation)
        rec = TreeRecorder()
        ep.iterate(max_iter=12, callback=rec, damping=e["damping"])
        tr = e["traj"]
        shape = tuple(e["shape"])
        for it, got in enumerate(rec.iters):
            for key, val in got.items():
                ref = np.asarray(tr[key][it], dtype=float)
                if key.endswith(":a") or key.startswith("v:"):
                    close(np.ravel(val)[0], ref, 1e-8)
                else:
                    sc = max(1.0, float(np.abs(ref).max()))
                    close(np.ravel(val), np.ravel(ref), 1e-7, 1e-8 * sc)
        assert ep.n_iter == 12


def test_tree_ep_ensemble_matches_oracle(ctx, golden_tree):
    "three instances (different observations) in one ensemble = three oracle runs; fp32 within 1e-3"
    import tramp_amd as T
    e = dict(golden_tree["ep"][0])
    shape = tuple(e["shape"])
    rs = np.random.RandomState(4)
    Y = np.stack([e["y"], e["y"] + 0.1 * rs.standard_normal(shape), rs.standard_normal(shape)])
    runs = []
    for b in range(3):
        eb = dict(e, y=Y[b])
        factors = dict(GaussianPrior=orc.GaussPriorFactor(shape), GaussianLikelihood=orc.GaussLikFactor(Y[b], e["noise_var"]),
                       GradientChannel=orc.GradientFactor(shape),
                       GaussBernoulliPrior=orc.GBPriorFactor((1,) + shape, rho=e["rho"]))
        runs.append(orc.ep_tree(factors, ["x", "z"], TREE_EDGES, dict(x=shape, z=(1,) + shape), max_iter=15,
                                damping=0.2, tol=0.0, max_increase=np.inf))
        del eb
    for dtype, rt in ((np.float64, 1e-8), (np.float32, 2e-3)):
        model = build_tree(T, ctx, e, y=Y, dtype=dtype)
        ep = T.ExpectationPropagation(model, ctx=ctx)
        assert ep.batch == 3
        ep.iterate(max_iter=15, callback=T.PassCallback(), damping=0.2)
        data = ep.get_variables_data()
        for b in range(3):
            close(data["x"]["r"][b], runs[b]["r"]["x"].ravel(), rt, rt)
            close(data["z"]["r"][b], runs[b]["r"]["z"].ravel(), rt, rt)
            close(data["x"]["v"][b], runs[b]["v"]["x"], max(rt, 1e-7))
            close(data["z"]["v"][b], runs[b]["v"]["z"], max(rt, 1e-7))


def test_tree_ep_per_instance_stopping_on_the_device(ctx, golden_tree):
    """The tree scheduler's stopping rule and NaN guard run on the device as the chain's (csrc/monitor.hip): each
    instance of an ensemble stops at the iteration its own run would (MessagePassing.iterate,
    message_passing.py:345-356 with EarlyStoppingEP, callbacks.py:250-286) and reports that iterate; polling every
    4th iteration changes nothing; every message of a stopped instance is the one of its stopping iteration."""
    import tramp_amd as T
    e = dict(golden_tree["ep"][0])
    shape = tuple(e["shape"])
    rs = np.random.RandomState(4)
    Y = np.stack([e["y"], e["y"] + 0.3 * rs.standard_normal(shape), 2.0 * rs.standard_normal(shape),
                  0.05 * rs.standard_normal(shape)])
    B = Y.shape[0]
    rule = dict(tol=3e-4, wait_increase=5, max_increase=0.2)
    runs = []
    for b in range(B):
        factors = dict(GaussianPrior=orc.GaussPriorFactor(shape), GaussianLikelihood=orc.GaussLikFactor(Y[b], e["noise_var"]),
                       GradientChannel=orc.GradientFactor(shape),
                       GaussBernoulliPrior=orc.GBPriorFactor((1,) + shape, rho=e["rho"]))
        runs.append(orc.ep_tree(factors, ["x", "z"], TREE_EDGES, dict(x=shape, z=(1,) + shape), max_iter=200,
                                damping=0.2, **rule))
    assert len({r["n_iter"] for r in runs}) > 1                       # the instances stop at different iterations
    outs = []
    for kw in (dict(), dict(check_every=4)):
        model = build_tree(T, ctx, e, y=Y)
        ep = T.ExpectationPropagation(model, ctx=ctx)
        assert type(ep).__name__ == "TreeExpectationPropagation" and ep._device_monitor
        ep.iterate(max_iter=200, callback=T.EarlyStoppingEP(**rule), damping=0.2, **kw)
        assert ep._restore
        data = ep.get_variables_data()
        for b in range(B):
            assert ep.n_iter_instance[b] == runs[b]["n_iter"], (b, ep.n_iter_instance, [r["n_iter"] for r in runs])
            assert {1: "converged", 2: "diverged", 0: "max_iter"}[int(ep.status[b])] == runs[b]["status"]
            close(data["x"]["r"][b], runs[b]["r"]["x"].ravel(), 1e-7, 1e-8)
            close(data["z"]["r"][b], runs[b]["r"]["z"].ravel(), 1e-7, 1e-8)
            close(data["x"]["v"][b], runs[b]["v"]["x"], 1e-7)
        name = lambda n: getattr(n, "id", None) or type(n).__name__                     # noqa: E731
        outs.append((ep.n_iter_instance.copy(), data["x"]["r"].copy(),
                     {(name(s_), name(t_)): ep.get_message(s_, t_)[1].copy() for (s_, t_) in ep.msg}))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert np.array_equal(outs[0][2][k], outs[1][2][k])


def test_tree_ep_example_setting_matches_oracle(ctx):
    """The reference example's run (examples/sparse/plot_sparse_gradient.py:63,88-91: rho = 0.04, noise 1e-2,
    damping 0.1, EarlyStoppingEP(tol=1e-2)) at n = 4096: same stopping iteration and estimate as the oracle,
    and the estimate improves on the raw observation."""
    import tramp_amd as T
    rs = np.random.RandomState(1)
    n, rho, nv = 4096, 0.04, 1e-2
    jumps = rs.standard_normal(n) * rs.binomial(1, rho, n)
    x = np.cumsum(jumps)
    x -= x.mean()
    y = x + np.sqrt(nv) * rs.standard_normal(n)
    factors = dict(GaussianPrior=orc.GaussPriorFactor((n,)), GaussianLikelihood=orc.GaussLikFactor(y, nv),
                   GradientChannel=orc.GradientFactor((n,)), GaussBernoulliPrior=orc.GBPriorFactor((1, n), rho=rho))
    want = orc.ep_tree(factors, ["x", "z"], TREE_EDGES, dict(x=(n,), z=(1, n)), max_iter=1000, damping=0.1, tol=1e-2)
    model = build_tree(T, ctx, dict(shape=(n,), noise_var=nv, rho=rho), y=y)
    ep = T.ExpectationPropagation(model, ctx=ctx).iterate(max_iter=1000, damping=0.1,
                                                          callback=T.EarlyStoppingEP(tol=1e-2))
    assert ep.n_iter == want["n_iter"]
    r = ep.get_variable_data("x")["r"]
    close(r, want["r"]["x"], 1e-7, 1e-8)
    close(ep.get_variable_data("z")["r"], want["r"]["z"].ravel(), 1e-7, 1e-8)
    assert np.mean((r - x) ** 2) < np.mean((y - x) ** 2)


def test_tree_regression_matches_reference(ctx, golden_tree):
    """Sparse-gradient regression (total_variation_model.py:67-71): the gradient tree feeding a dense
    LinearChannel - every message and posterior per iteration, and the stopping iteration."""
    import tramp_amd as T
    from tramp_amd.base import Variable
    for e in golden_tree["reg"]:
        model = T.sparse_gradient_regression(e["A"], e["y"], tuple(e["shape"]), e["rho"], e["noise_var"], ctx=ctx)
        assert not model.is_chain and [v.id for v in model.variables if v.n_prev == 2] == ["x'"]
        ep = T.ExpectationPropagation(model, ctx=ctx)
        iters = []

        class Rec(T.EarlyStoppingEP):
            def __call__(self, algo, i, max_iter):
                rec = {}
                for (src, dst) in algo.msg:
                    key = f"{src.id if isinstance(src, Variable) else type(src).__name__}->" \
                          f"{dst.id if isinstance(dst, Variable) else type(dst).__name__}"
                    rec[key + ":a"], rec[key + ":b"] = algo.get_message(src, dst)     # z is carried rotated
                for vid, d in algo.get_variables_data().items():
                    rec["r:" + vid], rec["v:" + vid] = np.array(d["r"]), np.array(d["v"])
                iters.append(rec)
                return super().__call__(algo, i, max_iter)
        ep.iterate(max_iter=30, callback=Rec(), damping=e["damping"])
        assert ep.n_iter == e["n_iter"] and [ep.variables[i].id for i in ep._rot] == ["z"]
        tr = e["traj"]
        for it, got in enumerate(iters[:len(tr["v:x"])]):
            for key, val in got.items():
                ref = np.asarray(tr[key][it], dtype=float)
                if key.endswith(":a") or key.startswith("v:"):
                    close(np.ravel(val)[0], ref, 1e-7)
                else:
                    close(np.ravel(val), np.ravel(ref), 1e-6, 1e-7 * max(1.0, float(np.abs(ref).max())))


def test_reshape_channel(ctx):
    import tramp_amd as T
    ch = T.ReshapeChannel(prev_shape=(3, 4), next_shape=12, ctx=ctx)
    b = np.arange(12.0).reshape(3, 4)
    a, out = ch.compute_forward_message(1.5, b, 0.0, np.zeros(12))
    assert a == 1.5 and out.shape == (12,) and np.array_equal(out, b.ravel())
    a, out = ch.compute_backward_message(0.0, np.zeros((3, 4)), 2.5, np.arange(12.0))
    assert a == 2.5 and out.shape == (3, 4) and np.array_equal(out, b)
    with pytest.raises(ValueError):
        T.ReshapeChannel(prev_shape=(3, 4), next_shape=11, ctx=ctx)


def test_tree_ep_graph_replay_is_identical(ctx, golden_tree):
    "iterate(graph=True) replays the captured launch sequence of a tree iteration: same bits as the eager loop"
    import tramp_amd as T
    e = golden_tree["reg"][1]
    outs = []
    for graph in (False, True):
        model = T.sparse_gradient_regression(e["A"], e["y"], tuple(e["shape"]), e["rho"], e["noise_var"], ctx=ctx)
        ep = T.ExpectationPropagation(model, ctx=ctx)
        ep.iterate(max_iter=9, callback=T.PassCallback(), damping=0.2, graph=graph)
        outs.append({k: (np.array(v["r"]), np.array(v["v"])) for k, v in ep.get_variables_data().items()})
    for k in outs[0]:
        assert np.array_equal(outs[0][k][0], outs[1][k][0]) and np.array_equal(outs[0][k][1], outs[1][k][1]), k


def test_gauss_channel_hidden_factor_matches_reference(ctx, golden_tree):
    "GaussianChannel between two variables (noisy 1-bit CS): messages and the EP trajectory of the reference"
    import tramp_amd as T
    for c in golden_tree["gch"]:
        N, M = c["N"], int(c["alpha"] * c["N"])
        W = np.random.RandomState(c["seed"]).randn(M, N) / np.sqrt(N)
        g = T.GaussianChannel(var=c["var"], ctx=ctx)
        bz = c["msgs"]["bz"]
        a, b = g.compute_forward_message(1.3, bz, 0.7, 2 * bz)
        close(a, c["msgs"]["fwd"][0], 1e-14)
        close(b, c["msgs"]["fwd"][1], 1e-14)
        a, b = g.compute_backward_message(1.3, bz, 0.7, 2 * bz)
        close(a, c["msgs"]["bwd"][0], 1e-14)
        close(b, c["msgs"]["bwd"][1], 1e-14)
        model = (T.GaussBernoulliPrior(size=N, rho=0.3, ctx=ctx) @ T.SISOVariable(id="x")
                 @ T.LinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z") @ g @ T.SISOVariable(id="a")
                 @ T.SgnLikelihood(y=c["y"], ctx=ctx)).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx)
        iters = []

        class Rec(T.EarlyStoppingEP):
            def __call__(self, algo, i, max_iter):
                iters.append({k: (np.array(v["r"]), float(v["v"])) for k, v in algo.get_variables_data().items()})
                return super().__call__(algo, i, max_iter)
        ep.iterate(max_iter=40, callback=Rec(), damping=c["damping"])
        assert ep.n_iter == c["n_iter"]
        t = c["traj"]
        for it in range(t["v:x"].shape[0]):
            for vid in ("x", "z", "a"):
                close(iters[it][vid][1], t[f"v:{vid}"][it], 1e-6)
                close(iters[it][vid][0], t[f"r:{vid}"][it], 1e-6, 1e-6 * np.abs(t[f"r:{vid}"][it]).max())
        close(iters[-1]["x"][0], c["final_r"], 1e-4, 1e-5)


def test_tree_state_evolution_matches_reference(ctx, golden_tree):
    "StateEvolution on a tree model dispatches to the tree recursion; messages and variances per iteration"
    import tramp_amd as T
    from tramp_amd.base import Variable
    from tramp_amd.tree import TreeStateEvolution
    from tramp_amd.variables import SIMOVariable, MILeafVariable
    for c in golden_tree["se"]:
        shape = tuple(c["shape"])
        d = len(shape)
        model = (T.GaussianPrior(size=shape, ctx=ctx) @ SIMOVariable(id="x", n_next=2) @ (
            T.GaussianLikelihood(y=None, var=c["noise_var"], ctx=ctx) + (
                T.GradientChannel(shape=shape, ctx=ctx) + T.GaussBernoulliPrior(size=(d,) + shape, rho=c["rho"], ctx=ctx)
            ) @ MILeafVariable(id="z", n_prev=2))).to_model()
        se = T.StateEvolution(model)
        assert isinstance(se, TreeStateEvolution)
        for k, tau in c["tau"].items():
            close(se.taus[k], tau, 1e-12)
        iters = []

        def cb(algo, i, max_iter):
            rec = {f"{s.id if isinstance(s, Variable) else type(s).__name__}->"
                   f"{t.id if isinstance(t, Variable) else type(t).__name__}:a": float(a) for (s, t), a in algo.msg.items()}
            rec.update({"v:" + k: float(v["v"]) for k, v in algo.get_variables_data().items()})
            iters.append(rec)
            return False
        se.iterate(max_iter=12, callback=cb, damping=c["damping"])
        t = c["traj"]
        for it, rec in enumerate(iters):
            for k, v in rec.items():
                close(v, t[k][it], 1e-6)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_l21_prior_matches_reference(ctx, golden_tree, dtype):
    "MAP_L21NormPrior (group soft-threshold along axis 0): posterior, message, exact support"
    import tramp_amd as T
    rt = 1e-12 if dtype == np.float64 else 2e-5
    for c in golden_tree["l21"]:
        size = tuple(c["size"])
        pr = T.MAP_L21NormPrior(size=size, gamma=c["gamma"], dtype=dtype, ctx=ctx)
        r, v = pr.compute_forward_posterior(c["ax"], c["bx"])
        assert r.shape == size
        close(r, c["r"], rt, rt)
        close(v, c["v"], max(rt, 1e-12))
        a, b = pr.compute_forward_message(c["ax"], c["bx"])
        close(a, c["a_new"], max(rt * 10, 1e-11))
        close(b, c["b_new"], rt * 10, rt * 10 * max(1.0, float(np.abs(c["b_new"]).max())))
        if dtype == np.float64:
            mask, count = pr.support(c["ax"], c["bx"])
            assert np.array_equal(mask, c["mask"]) and int(count[0]) == int(c["mask"].sum())
    # ensemble rows are independent; a zero group gives nan like the reference's 0 * nan
    c = golden_tree["l21"][0]
    pr = T.MAP_L21NormPrior(size=tuple(c["size"]), gamma=c["gamma"], ctx=ctx)
    r2, v2 = pr.compute_forward_posterior(np.array([c["ax"], 2 * c["ax"]]), np.stack([c["bx"], 0.5 * c["bx"]]))
    close(r2[0], c["r"], 1e-12, 1e-12)
    want_r, want_v = orc.l21_prior_posterior(2 * c["ax"], 0.5 * c["bx"], c["gamma"])
    close(r2[1], want_r, 1e-12, 1e-12)
    close(v2, [c["v"], want_v], 1e-12)
    bz = c["bx"].copy()
    bz[:, 0] = 0.0
    assert np.isnan(pr.compute_forward_posterior(c["ax"], bz)[1])
    with pytest.raises(NotImplementedError):
        T.MAP_L21NormPrior(size=(2, 4), axis=1, ctx=ctx)


def test_total_variation_ep_matches_reference(ctx, golden_tree):
    "TV denoising (gradient tree with the L21 prior): reference trajectories from a constant initialisation"
    import tramp_amd as T
    from tramp_amd.variables import SIMOVariable, MILeafVariable
    for e in golden_tree["tv"]:
        shape = tuple(e["shape"])
        d, n = len(shape), int(np.prod(shape))
        model = (T.GaussianPrior(size=shape, ctx=ctx) @ SIMOVariable(id="x", n_next=2) @ (
            T.GaussianLikelihood(y=e["y"].reshape(n), var=e["noise_var"], ctx=ctx) + (
                T.GradientChannel(shape=shape, ctx=ctx) + T.MAP_L21NormPrior(size=(d,) + shape, gamma=e["gamma"], ctx=ctx)
            ) @ MILeafVariable(id="z", n_prev=2))).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx)
        iters = []

        class Rec(T.EarlyStoppingEP):
            def __call__(self, algo, i, max_iter):
                iters.append({k: (np.array(v["r"]), float(v["v"])) for k, v in algo.get_variables_data().items()})
                return super().__call__(algo, i, max_iter)
        ep.iterate(max_iter=30, callback=Rec(), damping=e["damping"], initializer=T.ConstantInit(a=e["a0"], b=e["b0"]))
        assert ep.n_iter == e["n_iter"]
        tr = e["traj"]
        for it in range(len(tr["v:x"])):
            for vid in ("x", "z"):
                close(iters[it][vid][1], tr["v:" + vid][it], 1e-7)
                close(iters[it][vid][0], np.ravel(tr["r:" + vid][it]), 1e-6, 1e-7)
        if e["name"] == "tv2d":
            assert np.mean((np.ravel(iters[-1]["x"][0]) - e["x"].ravel()) ** 2) < np.mean((e["y"] - e["x"]) ** 2)


def test_generative_prior_chain_matches_reference(ctx, golden_tree):
    """VAE-like generative prior (examples/vae_prior/plot_vae.py): Gaussian latent -> Linear -> Bias -> LeakyRelu
    -> Linear -> Bias -> HardTanh -> Gaussian likelihood; BiasChannel messages and the EP trajectory."""
    import tramp_amd as T
    for c in golden_tree["vae"]:
        D, H, N = c["dims"]
        bch = T.BiasChannel(c["b1"], ctx=ctx)
        bb = c["msgs"]["b"]
        a, b = bch.compute_forward_message(1.3, bb, 0.7, 2 * bb)
        close(a, c["msgs"]["fwd"][0], 1e-14)
        close(b, c["msgs"]["fwd"][1], 1e-13, 1e-14)
        a, b = bch.compute_backward_message(1.3, bb, 0.7, 2 * bb)
        close(a, c["msgs"]["bwd"][0], 1e-14)
        close(b, c["msgs"]["bwd"][1], 1e-13, 1e-14)
        close(bch.second_moment(1.7), c["msgs"]["tau"], 1e-14)
        V = T.SISOVariable
        model = (T.GaussianPrior(size=D, ctx=ctx) @ V(id="z") @ T.LinearChannel(c["W1"], ctx=ctx, svd="host") @ V(id="a1")
                 @ T.BiasChannel(c["b1"], ctx=ctx) @ V(id="a2") @ T.LeakyReluChannel(slope=0.1, ctx=ctx) @ V(id="a3")
                 @ T.LinearChannel(c["W2"], ctx=ctx, svd="host") @ V(id="a4") @ T.BiasChannel(c["b2"], ctx=ctx)
                 @ V(id="a5") @ T.HardTanhChannel(ctx=ctx) @ V(id="x")
                 @ T.GaussianLikelihood(y=c["y"], var=c["noise_var"], ctx=ctx)).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx)
        iters = []

        def cb(algo, i, max_iter):
            iters.append({k: (np.array(v["r"]), float(v["v"])) for k, v in algo.get_variables_data().items()})
            return False
        ep.iterate(max_iter=40, callback=cb, damping=c["damping"])
        t = c["traj"]
        for it in range(t["v:x"].shape[0]):
            for vid in ("z", "a1", "a2", "a3", "a4", "a5", "x"):
                close(iters[it][vid][1], t[f"v:{vid}"][it], 1e-6)
                close(iters[it][vid][0], t[f"r:{vid}"][it], 1e-6, 1e-6 * max(1.0, float(np.abs(t[f"r:{vid}"][it]).max())))
        mse = float(np.mean((iters[-1]["x"][0] - c["x"]) ** 2))
        close(mse, c["mse"], 1e-4)
        assert mse < 0.3 * float(np.mean((c["y"] - c["x"]) ** 2))


def test_tree_to_observed_equals_direct_likelihood(ctx, golden_tree):
    "examples/sparse/plot_raccoon_denoise.py:111-119: `noise @ O('y')` on a tree, then to_observed({'y': y})"
    import tramp_amd as T
    e = golden_tree["ep"][2]
    shape = tuple(e["shape"])
    d = len(shape)
    generative = (T.GaussianPrior(size=shape, ctx=ctx) @ T.SIMOVariable(id="x", n_next=2) @ (
        T.GaussianChannel(var=e["noise_var"], ctx=ctx) @ T.SILeafVariable(id="y") + (
            T.GradientChannel(shape=shape, ctx=ctx) + T.GaussBernoulliPrior(size=(d,) + shape, rho=e["rho"], ctx=ctx)
        ) @ T.MILeafVariable(id="x'", n_prev=2))).to_model()
    assert [v.id for v in generative.variables if v.n_next == 0] == ["y", "x'"] or \
        sorted(v.id for v in generative.variables if v.n_next == 0) == ["x'", "y"]
    observed = generative.to_observed({"y": e["y"].ravel()})
    assert sorted(v.id for v in observed.variables) == ["x", "x'"]
    ep = T.ExpectationPropagation(observed, ctx=ctx).iterate(max_iter=12, callback=T.PassCallback())
    close(ep.get_variable_data("x")["r"], np.ravel(e["traj"]["r:x"][11]), 1e-7, 1e-8)
    close(ep.get_variable_data("x'")["r"], np.ravel(e["traj"]["r:z"][11]), 1e-7, 1e-8)


def test_committee_machine_matches_reference(ctx, golden_tree):
    """Committee machine (tramp/models/committee_model.py:8-41): K experts sharing the sensing matrix, summed by
    the K-input SumChannel.  SumChannel messages and every message / posterior of the reference's EP run."""
    import tramp_amd as T
    from tramp_amd.base import Variable
    for c in golden_tree["com"]:
        K, N, M = c["K"], c["N"], c["M"]
        m = c["msgs"]
        sm = T.SumChannel(n_prev=K, ctx=ctx)
        fa, fb = sm.compute_forward_message(m["az"], m["bz"], m["ax"], m["bx"])
        close(fa, m["fwd"][0], 1e-13)
        close(fb, m["fwd"][1], 1e-12, 1e-13)
        ba, bb = sm.compute_backward_message(m["az"], m["bz"], m["ax"], m["bx"])
        for k in range(K):
            close(ba[k], m["bwd"][0][k], 1e-13)
            close(bb[k], m["bwd"][1][k], 1e-12, 1e-13)
        F = np.random.RandomState(17 + K).standard_normal((M, N)) / np.sqrt(N)
        shared = T.LinearChannel(F, ctx=ctx, svd="host")
        experts = None
        for k in range(K):
            lin = shared if k == 0 else T.LinearChannel(F, ctx=ctx, factors=(shared._Uk, shared._sing, shared._Vt, shared.rank))
            e = (T.GaussianPrior(size=N, var=c["pvars"][k], ctx=ctx) @ T.SISOVariable(id=f"x_{k}") @ lin
                 @ T.SISOVariable(id=f"z_{k}") @ T.ReluChannel(ctx=ctx) @ T.SISOVariable(id=f"a_{k}"))
            experts = e if experts is None else experts + e
        model = (experts @ T.SumChannel(n_prev=K, ctx=ctx) @ T.SISOVariable(id="n")
                 @ T.GaussianLikelihood(y=c["y"], var=c["noise_var"], ctx=ctx)).to_model()
        assert not model.is_chain
        g = model.dag

        def name(node):
            if isinstance(node, Variable):
                return node.id
            ids = sorted(v.id for v in list(g.predecessors(node)) + list(g.successors(node)))
            return f"{type(node).__name__}[{','.join(ids)}]"
        ep = T.ExpectationPropagation(model, ctx=ctx)
        iters = []

        def cb(algo, i, max_iter):
            rec = {}
            for (src, dst), msg in algo.msg.items():
                rec[f"{name(src)}->{name(dst)}:a"], rec[f"{name(src)}->{name(dst)}:b"] = msg.a.get().copy(), msg.b.get().copy()
            for vid, d in algo.get_variables_data().items():
                rec["r:" + vid], rec["v:" + vid] = np.array(d["r"]), np.array(d["v"])
            iters.append(rec)
            return False
        ep.iterate(max_iter=12, callback=cb, damping=c["damping"])
        tr = c["traj"]
        for it, got in enumerate(iters):
            tol = 1e-7 if it < 6 else 1e-5
            for key, val in got.items():
                ref = np.asarray(tr[key][it], dtype=float)
                if key.endswith(":a") or key.startswith("v:"):
                    close(np.ravel(val)[0], ref, tol)
                else:
                    close(np.ravel(val), np.ravel(ref), tol, tol * max(1.0, float(np.abs(ref).max())))


def test_soft_committee_builder_end_to_end(ctx):
    "committee_model.py:57-73 through sample() -> to_observed() -> EP: the noisy sum is explained by the experts"
    import tramp_amd as T
    np.random.seed(3)
    model = T.soft_committee(K=2, N=32, alpha=3.0, ensemble_type="gaussian", prior_mean=0.0, prior_var=[1.0, 0.5],
                             noise_var=1e-2, ctx=ctx)
    X = model.sample()
    assert set(X) == {"x_0", "z_0", "a_0", "x_1", "z_1", "a_1", "n", "y"} and X["y"].shape == (96,)
    lins = [f for f in model.factors if isinstance(f, T.LinearChannel)]
    assert len(lins) == 2 and lins[0]._Vt.ptr == lins[1]._Vt.ptr          # one factorisation, shared
    observed = model.to_observed({"y": X["y"]})
    ep = T.ExpectationPropagation(observed, ctx=ctx).iterate(max_iter=50, damping=0.3)
    data = ep.get_variables_data()
    assert all(np.all(np.isfinite(d["r"])) for d in data.values())
    assert np.mean((data["n"]["r"] - X["n"]) ** 2) < 1e-2 * np.mean(X["n"] ** 2)
    assert np.mean((data["a_0"]["r"] + data["a_1"]["r"] - X["n"]) ** 2) < 1e-2 * np.mean(X["n"] ** 2)


def test_binary_prior_and_sgn_committee_match_reference(ctx, golden_tree):
    "BinaryPrior (binary_prior.py:8-62) and the sign committee machine (committee_model.py:44-54), K = 3"
    import tramp_amd as T
    from tramp_amd.base import Variable
    for c in golden_tree["binp"]:
        for dtype, rt in ((np.float64, 1e-13), (np.float32, 2e-6)):
            pr = T.BinaryPrior(size=9, p_pos=c["p_pos"], dtype=dtype, ctx=ctx)
            r, v = pr.compute_forward_posterior(1.3, c["bx"])
            close(r, c["r"], rt, rt)
            close(v, c["v"], max(rt, 1e-12) * 10)
            a, b = pr.compute_forward_message(1.3, c["bx"])
            close(a, c["a_new"], max(rt * 100, 1e-11))
            close(b, c["b_new"], rt * 100, rt * 100 * max(1.0, float(np.abs(c["b_new"]).max())))
    c = golden_tree["sgncom"][0]
    K, N, M = c["K"], c["N"], c["M"]
    F = np.random.RandomState(29).standard_normal((M, N)) / np.sqrt(N)
    shared = T.LinearChannel(F, ctx=ctx, svd="host")
    experts = None
    for k in range(K):
        lin = shared if k == 0 else T.LinearChannel(F, ctx=ctx, factors=(shared._Uk, shared._sing, shared._Vt, shared.rank))
        e = (T.BinaryPrior(size=N, p_pos=c["p_pos"][k], ctx=ctx) @ T.SISOVariable(id=f"x_{k}") @ lin
             @ T.SISOVariable(id=f"z_{k}") @ T.SgnChannel(ctx=ctx) @ T.SISOVariable(id=f"a_{k}"))
        experts = e if experts is None else experts + e
    model = (experts @ T.SumChannel(n_prev=K, ctx=ctx) @ T.SISOVariable(id="a") @ T.SgnLikelihood(y=c["y"], ctx=ctx)).to_model()
    g = model.dag

    def name(node):
        if isinstance(node, Variable):
            return node.id
        return f"{type(node).__name__}[{','.join(sorted(v.id for v in list(g.predecessors(node)) + list(g.successors(node))))}]"
    iters = []

    def cb(algo, i, max_iter):
        rec = {}
        for (src, dst), msg in algo.msg.items():
            rec[f"{name(src)}->{name(dst)}:a"], rec[f"{name(src)}->{name(dst)}:b"] = msg.a.get().copy(), msg.b.get().copy()
        for vid, d in algo.get_variables_data().items():
            rec["r:" + vid] = np.array(d["r"])
        iters.append(rec)
        return False
    T.ExpectationPropagation(model, ctx=ctx).iterate(max_iter=10, callback=cb, damping=c["damping"])
    tr = c["traj"]
    for it, got in enumerate(iters):
        tol = 1e-6 if it < 5 else 1e-4
        for key, val in got.items():
            ref = np.asarray(tr[key][it], dtype=float)
            if key.endswith(":a") and not key.startswith("r:"):
                close(np.ravel(val)[0], ref, tol)
            else:
                close(np.ravel(val), np.ravel(ref), tol, tol * max(1.0, float(np.abs(ref).max())))


def test_committee_state_evolution_matches_reference(ctx, golden_tree):
    "TreeStateEvolution through the K-input SumChannel: the reference's 3-minute dblquad run, here milliseconds"
    import tramp_amd as T
    from tramp_amd.base import Variable
    for c in golden_tree["comse"]:
        K = c["K"]
        experts = None
        for k in range(K):
            e = (T.GaussianPrior(size=None, var=c["pvars"][k], ctx=ctx) @ T.SISOVariable(id=f"x_{k}")
                 @ T.MarchenkoPasturChannel(alpha=c["alpha"]) @ T.SISOVariable(id=f"z_{k}") @ T.ReluChannel(ctx=ctx)
                 @ T.SISOVariable(id=f"a_{k}"))
            experts = e if experts is None else experts + e
        model = (experts @ T.SumChannel(n_prev=K, ctx=ctx) @ T.SISOVariable(id="n")
                 @ T.GaussianLikelihood(y=None, var=c["noise_var"], ctx=ctx)).to_model()
        se = T.StateEvolution(model)
        for k, tau in c["tau"].items():
            close(se.taus[k], tau, 1e-12)
        g = model.dag

        def name(node):
            if isinstance(node, Variable):
                return node.id
            return f"{type(node).__name__}[{','.join(sorted(v.id for v in list(g.predecessors(node)) + list(g.successors(node))))}]"
        iters = []

        def cb(algo, i, max_iter):
            rec = {f"{name(s)}->{name(t)}:a": float(a) for (s, t), a in algo.msg.items()}
            rec.update({"v:" + k: float(v["v"]) for k, v in algo.get_variables_data().items()})
            iters.append(rec)
            return False
        se.iterate(max_iter=6, callback=cb, damping=c["damping"])
        t = c["traj"]
        for it, rec in enumerate(iters):
            for k, v in rec.items():
                close(v, t[k][it], 1e-5)


def test_tracking_callbacks_on_a_tree(ctx, golden_tree):
    "TrackMessages / TrackEstimate / JoinCallback work on the tree scheduler (callbacks.py:33-62,111-134)"
    import tramp_amd as T
    e = golden_tree["ep"][0]
    model = build_tree(T, ctx, e)
    tm, te = T.TrackMessages(), T.TrackEstimate()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=3, callback=T.JoinCallback([tm, te, T.EarlyStoppingEP()]))
    dirs = {(r["id"], r["direction"]) for r in tm.records}
    assert ("x", "fwd") in dirs and any(d.startswith("fwd[") for v, d in dirs if v == "z") and len(tm.records) == 3 * 5
    got = [r for r in tm.records if r["id"] == "x" and r["direction"] == "fwd"]
    close([r["a"] for r in got], e["traj"]["GaussianPrior->x:a"][:3], 1e-9)
    assert len(te.records) == 3 * 2


def test_binary_prior_state_evolution_matches_reference(ctx, golden_tree):
    "BinaryPrior.compute_forward_error on the device quadrature (binary_prior.py:82-86)"
    import tramp_amd as T
    for c in golden_tree["binse"]:
        pr = T.BinaryPrior(size=None, p_pos=c["p_pos"], ctx=ctx)
        close(pr.compute_forward_error(c["ax"]), c["v"], 1e-7, 1e-12)          # the reference's quad stops at ~1e-8 abs
    pr = T.BinaryPrior(size=None, p_pos=0.5, ctx=ctx)
    v = pr.compute_forward_error(np.array([0.05, 0.7, 3.0]))
    assert v.shape == (3,) and np.all(np.diff(v) < 0)
    close(pr.compute_forward_state_evolution(0.7), 1 / pr.compute_forward_error(0.7) - 0.7, 1e-12)


@pytest.mark.parametrize("which", ["sparse_gradient_classification", "tv_regression", "tv_classification"])
def test_remaining_total_variation_models_match_reference(ctx, golden_models2, which):
    """The other three builders of tramp/models/total_variation_model.py (:66-84; the L21 ones are the model its docstrings
    describe - the reference's own tv_block raises on scale=): every message and posterior of every iteration of the reference's
    EP (tools/make_golden_models2.py built the same graphs from the reference's factor classes), and the stopping iteration."""
    import tramp_amd as T
    from tramp_amd.base import Variable
    for e in [m for m in golden_models2["models"] if m["name"] == which]:
        assert e["error"] is None
        model = getattr(T, which)(e["A"], e["y"], tuple(e["shape"]), noise_var=e["noise_var"], ctx=ctx, **e["kwargs"])
        assert not model.is_chain and [v.id for v in model.variables if v.n_prev == 2] == ["x'"]
        assert [v.id for v in model.variables] == (["x", "x'", "r", "z", "a"] if which.endswith("classification") else ["x", "x'", "r", "z"])
        ep = T.ExpectationPropagation(model, ctx=ctx)
        iters = []

        class Rec(T.EarlyStoppingEP):
            def __call__(self, algo, i, max_iter):
                rec = {}
                for (src, dst) in algo.msg:
                    key = f"{src.id if isinstance(src, Variable) else type(src).__name__}->" \
                          f"{dst.id if isinstance(dst, Variable) else type(dst).__name__}"
                    rec[key + ":a"], rec[key + ":b"] = algo.get_message(src, dst)     # z is carried rotated
                for vid, d in algo.get_variables_data().items():
                    rec["r:" + vid], rec["v:" + vid] = np.array(d["r"]), np.array(d["v"])
                iters.append(rec)
                return super().__call__(algo, i, max_iter)
        ep.iterate(max_iter=30, callback=Rec(), damping=e["damping"], initializer=T.ConstantInit(a=e["a0"], b=e["b0"]))
        assert ep.n_iter == e["n_iter"], (which, e["shape"])
        tr = e["traj"]
        assert set(k for k in tr if "->" in k) == set(k for k in iters[0] if "->" in k)
        for it, got in enumerate(iters[:len(tr["v:x"])]):
            # the reference forms a variable -> factor precision as (sum of the incoming ones) - (that factor's own): with the
            # L21 prior's message at the 7e10 clip the other parent's 0.62 comes back as 0.619415283203125 (ulp of the sum:
            # 1.5e-5); here it is the other parent's message itself
            a_max = max(float(tr[k][it]) for k in tr if "->" in k and k.endswith(":a"))
            for key, val in got.items():
                ref = np.asarray(tr[key][it], dtype=float)
                if ("->" in key and key.endswith(":a")) or key.startswith("v:"):     # (the variable "a" has the key "r:a")
                    close(np.ravel(val)[0], ref, 1e-6, 4e-16 * a_max if "->" in key else 0.0)
                else:
                    close(np.ravel(val), np.ravel(ref), 1e-5, 1e-6 * max(1.0, float(np.abs(ref).max())))
        close(iters[-1]["r:x"], np.ravel(e["final_r"]), 1e-4, 1e-5 * max(1.0, float(np.abs(e["final_r"]).max())))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_duplicate_channel_matches_reference(ctx, golden_models2, dtype):
    "DuplicateChannel (channels/shape/duplicate_channel.py): posterior, the backward message, the K forward messages; damping; ensembles"
    import tramp_amd as T
    rt = 1e-12 if dtype == np.float64 else 3e-6
    for c in golden_models2["dup_fn"]:
        K = c["K"]
        ch = T.DuplicateChannel(n_next=K, dtype=dtype, ctx=ctx)
        sc = max(1.0, float(np.abs(c["bz"]).max()))
        a, b = ch.compute_backward_message(c["az"], c["bz"], c["ax"], c["bx"])
        close(a, c["a_bwd"], max(rt, 1e-12))
        close(b, c["b_bwd"], rt * 10, rt * 10 * sc)
        fa, fb = ch.compute_forward_message(c["az"], c["bz"], c["ax"], c["bx"])
        for k in range(K):
            close(fa[k], c["a_fwd"][k], max(rt, 1e-12))
            close(fb[k], c["b_fwd"][k], rt * 10, rt * 10 * sc)
            a1, b1 = ch.compute_forward_message(c["az"], c["bz"], c["ax"], c["bx"], target=k)
            assert np.array_equal(b1, fb[k]) and a1 == fa[k]
        r, v = ch.compute_backward_posterior(c["az"], c["bz"], c["ax"], c["bx"])
        close(r, c["rz"], rt * 10, rt * 10)
        close(v, c["vz"], max(rt, 1e-12))
        assert ch.compute_backward_error(c["az"], c["ax"], 1.0) == pytest.approx(c["v_bwd_err"], rel=1e-14)
        assert ch.second_moment(0.7) == (0.7,) * K and len(ch.sample(c["bz"])) == K
    # an ensemble of instances with per-instance precisions, and fused damping, against the oracle
    rs = np.random.RandomState(5)
    B, n, K = 3, 257, 3
    ch = T.DuplicateChannel(n_next=K, dtype=dtype, ctx=ctx)
    az, bz = rs.uniform(0.5, 2, B), rs.standard_normal((B, n))
    ax, bx = [rs.uniform(0.2, 3, B) for _ in range(K)], [rs.standard_normal((B, n)) for _ in range(K)]
    a, b = ch.compute_backward_message(az, bz, ax, bx)
    for i in range(B):
        wa, wb = orc.duplicate_backward_message(az[i], bz[i], [q[i] for q in ax], [q[i] for q in bx])
        close(a[i], wa, max(rt, 1e-12))
        close(b[i], wb, rt * 10, rt * 40)
    if dtype == np.float64:
        old_a, old_b = ctx.array(rs.uniform(0.5, 1, B), np.float64), ctx.array(rs.standard_normal((B, n)), np.float64)
        oa, ob = old_a.get(), old_b.get()
        out_a, out_b = ctx.empty((B,), np.float64), ctx.empty((B, n), np.float64)
        ch.compute_forward_message(ctx.array(az, np.float64), ctx.array(bz, np.float64), [ctx.array(q, np.float64) for q in ax],
                                   [ctx.array(q, np.float64) for q in bx], target=1, damping=0.3, old_a=old_a, old_b=old_b,
                                   out_a=out_a, out_b=out_b)
        for i in range(B):
            fa, fb = orc.duplicate_forward_message(az[i], bz[i], [q[i] for q in ax], [q[i] for q in bx])
            close(out_a.get()[i], 0.3 * oa[i] + 0.7 * fa[1], 1e-12)
            close(out_b.get()[i], 0.3 * ob[i] + 0.7 * fb[1], 1e-11, 1e-12)
    with pytest.raises(ValueError):
        ch.compute_backward_message(1.0, bz[0], ax[:2], bx[:2])


def test_duplicate_channel_ep_matches_reference(ctx, golden_models2):
    """A signal observed twice - through Gaussian noise and through a dense matrix - behind a DuplicateChannel (the one factor with
    several outputs): every message and posterior of every iteration of the reference's EP, the stopping iteration, and the
    same model with the SIMO variable that does the same job."""
    import tramp_amd as T
    from tramp_amd.base import Variable
    from tramp_amd.variables import SIMOVariable
    for e in golden_models2["dup_ep"]:
        N = e["N"]
        model = (T.GaussBernoulliPrior(size=N, rho=e["rho"], ctx=ctx) @ T.SISOVariable(id="z") @ T.DuplicateChannel(n_next=2, ctx=ctx) @ (
            T.SISOVariable(id="x1") @ T.GaussianLikelihood(y=e["y1"], var=e["nv1"], ctx=ctx)
            + T.SISOVariable(id="x2") @ T.LinearChannel(e["A"], name="A", ctx=ctx) @ T.SISOVariable(id="w")
            @ T.GaussianLikelihood(y=e["y2"], var=e["nv2"], ctx=ctx))).to_model()
        assert not model.is_chain
        ep = T.ExpectationPropagation(model, ctx=ctx)
        iters = []

        class Rec(T.EarlyStoppingEP):
            def __call__(self, algo, i, max_iter):
                rec = {}
                for (src, dst) in algo.msg:
                    key = f"{src.id if isinstance(src, Variable) else type(src).__name__}->" \
                          f"{dst.id if isinstance(dst, Variable) else type(dst).__name__}"
                    rec[key + ":a"], rec[key + ":b"] = algo.get_message(src, dst)
                for vid, d in algo.get_variables_data().items():
                    rec["r:" + vid], rec["v:" + vid] = np.array(d["r"]), np.array(d["v"])
                iters.append(rec)
                return super().__call__(algo, i, max_iter)
        ep.iterate(max_iter=30, callback=Rec(), damping=e["damping"])
        assert ep.n_iter == e["n_iter"]
        tr = e["traj"]
        # two GaussianLikelihood factors share a class name in the recorder's keys: compare the keys that are unambiguous
        amb = [k for k in tr if "GaussianLikelihood" in k]
        for it, got in enumerate(iters[:len(tr["v:z"])]):
            for key, val in got.items():
                if key in amb:
                    continue
                ref = np.asarray(tr[key][it], dtype=float)
                if ("->" in key and key.endswith(":a")) or key.startswith("v:"):
                    close(np.ravel(val)[0], ref, 1e-7)
                else:
                    close(np.ravel(val), np.ravel(ref), 1e-6, 1e-7 * max(1.0, float(np.abs(ref).max())))
        close(iters[-1]["r:z"], e["final_r"], 1e-5, 1e-6)
        assert np.mean((iters[-1]["r:z"] - e["x"]) ** 2) < np.mean((e["y1"] - e["x"]) ** 2)
        # the same posterior from a variable with two children (what the reference's docs recommend instead)
        m2 = (T.GaussBernoulliPrior(size=N, rho=e["rho"], ctx=ctx) @ SIMOVariable(id="z", n_next=2) @ (
            T.GaussianLikelihood(y=e["y1"], var=e["nv1"], ctx=ctx)
            + T.LinearChannel(e["A"], name="A", ctx=ctx) @ T.SISOVariable(id="w")
            @ T.GaussianLikelihood(y=e["y2"], var=e["nv2"], ctx=ctx))).to_model()
        ep2 = T.ExpectationPropagation(m2, ctx=ctx).iterate(max_iter=200, damping=e["damping"], callback=T.EarlyStoppingEP(tol=1e-10))
        ep1 = T.ExpectationPropagation(model, ctx=ctx).iterate(max_iter=200, damping=e["damping"], callback=T.EarlyStoppingEP(tol=1e-10))
        close(ep1.get_variable_data("z")["r"], ep2.get_variable_data("z")["r"], 1e-5, 1e-6)


def test_duplicate_channel_state_evolution(ctx):
    "SE through the K-output factor: at the fixed point the precision into z is the sum of the outputs' (duplicate_channel.py:39-42)"
    import tramp_amd as T
    model = (T.GaussBernoulliPrior(size=100, rho=0.2, ctx=ctx) @ T.SISOVariable(id="z") @ T.DuplicateChannel(n_next=2, ctx=ctx) @ (
        T.SISOVariable(id="x1") @ T.GaussianLikelihood(y=None, var=0.5, ctx=ctx)
        + T.SISOVariable(id="x2") @ T.GaussianLikelihood(y=None, var=0.25, ctx=ctx))).to_model()
    se = T.StateEvolution(model)
    se.iterate(max_iter=50)
    data = se.get_variables_data()
    # two Gaussian observations of precisions 2 and 4 are one of precision 6
    ref = (T.GaussBernoulliPrior(size=100, rho=0.2, ctx=ctx) @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=None, var=1 / 6.0, ctx=ctx)).to_model()
    se2 = T.StateEvolution(ref)
    se2.iterate(max_iter=50)
    close(data["z"]["v"], se2.get_variables_data()["z"]["v"], 1e-9)
    close(data["x1"]["v"], data["z"]["v"], 1e-9)
