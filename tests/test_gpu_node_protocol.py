"""The drop-in claim at the level the reference's iterate() works on: tramp_amd factors driven ONLY
through node.forward_message(message) / node.backward_message(message) with
(source, target, dict(a=, b=, direction=)) triples, strict NumPy mode (host arrays in, host arrays out).

The loop below is this repository's own restatement, for a chain, of what the reference's scheduler
does around those two calls (tramp/algos/message_passing.py:211-232 init, :249-263 sweeps in
topological order and its reverse, :108-127 constant damping on factor->variable edges, :265-269 +
tramp/base.py:152-161 variable posterior; SISO variables relay messages, tramp/variables/
sub_variables.py:21-43; Factor.forward_message / backward_message, tramp/base.py:329-365).
Checked against the trajectories captured from the reference (tests/golden/ep_traj.npz): every
message and every posterior of the first iterations, for a sparse GLM with each output type and
for the two-layer ReLU model.
"""
import numpy as np
import pytest

from tests.test_gpu_ep import build_glm, build_two_layer

pytestmark = pytest.mark.gpu


class Relay:
    "a SISO variable as the protocol sees it: a node that passes messages through"

    def __init__(self, vid):
        self.id = vid


def protocol_ep(model, n_iter, damping):
    shapes = model.init_shapes()
    variables = list(model.variables)
    nodes = []                                   # topological order: f0, x0, f1, x1, ..., fL
    relay = {v.id: Relay(v.id) for v in variables}
    for v in variables:
        nodes += [model.prev_factor(v), relay[v.id]]
    nodes.append(model.next_factor(variables[-1]))
    # edges: (source, target) -> data; ConstantInit(a=0, b=0) on every edge in both directions
    edges = {}
    for v in variables:
        fp, fn, x = model.prev_factor(v), model.next_factor(v), relay[v.id]
        n = int(np.prod(shapes[v.id]))
        for s, t, d in ((fp, x, "fwd"), (x, fn, "fwd"), (fn, x, "bwd"), (x, fp, "bwd")):
            edges[(s, t)] = dict(a=0.0, b=np.zeros(n), direction=d)
    history = []

    def incoming(node):
        return [(s, t, d) for (s, t), d in edges.items() if t is node]

    def emit(node, new):
        for s, t, d in new:
            if damping and not isinstance(s, Relay):                     # factor -> variable edges only
                old = edges[(s, t)]
                d = dict(a=damping * old["a"] + (1 - damping) * d["a"],
                         b=damping * old["b"] + (1 - damping) * d["b"], direction=d["direction"])
            edges[(s, t)] = d

    def relay_messages(x, direction):
        # sub_variables.py:21-43: the fwd message of the previous factor goes on to the next one, and back
        out = []
        for s, t, d in incoming(x):
            if d["direction"] == direction:
                tgt = [tt for (ss, tt), dd in edges.items() if ss is x and dd["direction"] == direction][0]
                out.append((x, tgt, dict(a=d["a"], b=d["b"], direction=direction)))
        return out

    for it in range(n_iter):
        for node in nodes:                                               # forward sweep
            emit(node, relay_messages(node, "fwd") if isinstance(node, Relay) else node.forward_message(incoming(node)))
        for node in reversed(nodes):                                     # backward sweep
            emit(node, relay_messages(node, "bwd") if isinstance(node, Relay) else node.backward_message(incoming(node)))
        snap = {}
        for v in variables:                                              # base.py:152-161
            x, fp, fn = relay[v.id], model.prev_factor(v), model.next_factor(v)
            f, b = edges[(fp, x)], edges[(fn, x)]
            a = f["a"] + b["a"]
            snap[v.id] = dict(fwd=(f["a"], f["b"].copy()), bwd=(b["a"], b["b"].copy()),
                              r=(f["b"] + b["b"]) / a, v=1.0 / a)
        history.append(snap)
    return history


def _close(a, b, rtol, scale=None):
    b = np.asarray(b, dtype=np.float64)
    sc = np.abs(b).max() if scale is None else scale
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), b, rtol=rtol, atol=rtol * max(sc, 1e-300))


def _cases(golden_ep):
    seen, out = set(), []
    for c in golden_ep:
        key = (c["kind"], c.get("output"), c.get("damping"))
        if c["N"] in (64,) and key not in seen:
            seen.add(key)
            out.append(c)
    return out


def test_node_protocol_reproduces_reference_trajectories(ctx, golden_ep):
    import tramp_amd as T
    cases = _cases(golden_ep)
    assert {c["kind"] for c in cases} == {"glm", "two_layer"}
    for c in cases:
        model, ids = build_glm(T, ctx, c) if c["kind"] == "glm" else build_two_layer(T, ctx, c)
        n_it = 6
        hist = protocol_ep(model, n_it, c.get("damping"))
        t = c["traj"]
        for it in range(n_it):
            for vid in ids:
                s = hist[it][vid]
                a_tot = t[f"fwd:{vid}:a"][it] + t[f"bwd:{vid}:a"][it]
                _close(s["fwd"][0], t[f"fwd:{vid}:a"][it], 1e-7, scale=a_tot)
                _close(s["bwd"][0], t[f"bwd:{vid}:a"][it], 1e-7, scale=a_tot)
                _close(s["fwd"][1], t[f"fwd:{vid}:b"][it], 1e-7)
                _close(s["bwd"][1], t[f"bwd:{vid}:b"][it], 1e-7)
                _close(s["r"], t[f"r:{vid}"][it], 1e-7)
                _close(s["v"], t[f"v:{vid}"][it], 1e-7)


def test_node_protocol_returns_host_types(ctx):
    "strict mode: scalars come back as Python floats, vectors as NumPy arrays of the input's shape"
    import tramp_amd as T
    rs = np.random.RandomState(0)
    N, M = 32, 16
    prior = T.GaussBernoulliPrior(size=N, rho=0.3, ctx=ctx)
    lin = T.LinearChannel(rs.randn(M, N) / np.sqrt(N), ctx=ctx, svd="host")
    lik = T.SgnLikelihood(y=np.sign(rs.randn(M)), ctx=ctx)
    x, z = Relay("x"), Relay("z")
    bwd_x = dict(a=0.7, b=rs.randn(N), direction="bwd")
    (src, tgt, d), = prior.forward_message([(x, prior, bwd_x)])
    assert src is prior and tgt is x and d["direction"] == "fwd"
    assert isinstance(d["a"], float) and isinstance(d["b"], np.ndarray) and d["b"].shape == (N,)
    fwd_x = (x, lin, dict(a=d["a"], b=d["b"], direction="fwd"))
    bwd_z = (z, lin, dict(a=0.4, b=rs.randn(M), direction="bwd"))
    (s1, t1, d1), = lin.forward_message([fwd_x, bwd_z])
    (s2, t2, d2), = lin.backward_message([fwd_x, bwd_z])
    assert (s1, t1, d1["direction"], d1["b"].shape) == (lin, z, "fwd", (M,))
    assert (s2, t2, d2["direction"], d2["b"].shape) == (lin, x, "bwd", (N,))
    (s3, t3, d3), = lik.backward_message([(z, lik, dict(a=d1["a"], b=d1["b"], direction="fwd"))])
    assert s3 is lik and t3 is z and isinstance(d3["a"], float) and d3["b"].shape == (M,)
    assert prior.backward_message([(x, prior, bwd_x)]) == [] and lik.forward_message([]) == []
