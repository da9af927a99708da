"""Expectation propagation on tree-structured models (variables with several incoming or outgoing
factors): the schedule of tramp/algos/message_passing.py:249-269 with the variable nodes'
messages made explicit.

On a chain a SISO variable just relays its two factors' messages, which is what
`ExpectationPropagation` exploits (no copies).  On a tree a variable of degree k sends each
neighbour the sum of its incoming messages minus that neighbour's own
(`Variable.forward_message / backward_message`, tramp/base.py:186-211), *at the moment the sweep
visits the variable*; a factor later reads that stored value, not the live one (for a leaf with two
parents this matters: each parent sees the other's message as of the variable's last visit).  So
every factor-variable adjacency carries two buffers here, factor->variable and variable->factor,
and the sweeps visit all nodes in topological order, variables included.

Factor updates are the same device calls as on a chain; the variable->factor messages are one
`tramp_message_sum` launch (or a device copy when the variable has two neighbours).
"""
import ctypes as C

import numpy as np

from .algos import ExpectationPropagation, fill_message
from .base import Factor, Variable
from .device import default_context, dtype_code
from .likelihoods import GaussianLikelihood
from .priors import GaussianPrior


class _Msg:
    __slots__ = ("a", "b")

    def __init__(self, a, b):
        self.a, self.b = a, b


class TreeExpectationPropagation(ExpectationPropagation):
    # stopping rule, NaN guard and the per-instance freeze / restore run on the device as on a chain (csrc/monitor.hip:
    # the precision table simply has one row per directed factor<->variable message); TRAMP_TREE_MONITOR=host keeps the
    # host-side evaluation on the scalar tables
    import os as _os
    _device_monitor = _os.environ.get("TRAMP_TREE_MONITOR", "device") != "host"
    # restore_messages: as on the chain the snapshot is not a copy - two sets of message objects alternate (self.msg is the
    # set this iteration writes, self._msg_prev the one the previous iteration wrote).  A message not yet rewritten in
    # this iteration is read from the previous set (_latest), its value before the update (for damping) too (_old).
    _pingpong = True

    """Gaussian EP on a tree of factors and variables (tramp/algos/expectation_propagation.py:5-32)."""

    def __init__(self, model, batch=None, ctx=None):
        self.model = model
        self.shapes = model.init_shapes()
        self.variables = model.variables
        self.factors = model.factors
        for f in self.factors:
            if f.n_next > 1 and not (f.n_prev == 1 and hasattr(f, "_message")):
                raise ValueError(f"{f}: factors with several outputs are outside the HIP path (DuplicateChannel is the one there is)")
        for v in self.variables:
            if v.n_prev == 0:
                raise ValueError(f"variable {v.id} has no incoming factor")
            if v.n_prev + v.n_next < 2:
                raise ValueError(f"variable {v.id} is attached to a single factor; call model.to_observed(...) "
                                 "for observed leaves")
        dev_factors = [f for f in self.factors if hasattr(f, "_dt")]
        dts = {f.dtype for f in dev_factors}
        if len(dts) != 1:
            raise ValueError(f"all factors must compute in the same dtype, got {dts}")
        self.dtype = dts.pop()
        self.ctx = ctx or next((f._ctx for f in dev_factors if f._ctx is not None), None) or default_context()
        for f in dev_factors:
            f._ctx = self.ctx
        for f in self.factors:                                  # observations seen as [instances, components] of their variable
            if hasattr(f, "bind_shape") and f.n_prev == 1:
                (var,) = self.model.dag.predecessors(f)
                f.bind_shape(self.shapes.get(var.id))
        if batch is None:
            batch = 1
            for f in self.factors:
                yd = getattr(f, "_y_dev", None)
                if yd is not None:
                    batch = yd.shape[0]
        self.batch = int(batch)
        self.n_iter = 0
        self.n_iter_instance = np.zeros(self.batch, dtype=int)
        self.status = np.zeros(self.batch, dtype=int)
        self._allocated = False
        self._find_tree_rotations()
        from .algos import EarlyStoppingEP
        self.default_stopping = EarlyStoppingEP()

    def _find_tree_rotations(self):
        """The rotated OUTPUT basis of the chain scheduler (algos.py `_find_rotations`) for a variable that sits
        between a dense LinearChannel (M <= N) and an isotropic GaussianLikelihood, e.g. the measurement branch
        of sparse-gradient regression.  All four messages around that variable live in the rotated basis."""
        import os
        from .channels import LinearChannel
        self._rot = {}
        if os.environ.get("TRAMP_ROTATE_OUTPUT", "1") == "0":
            return
        g = self.model.dag
        for i, v in enumerate(self.variables):
            preds, succs = list(g.predecessors(v)), list(g.successors(v))
            if len(preds) == 1 and len(succs) == 1 and type(preds[0]) is LinearChannel \
                    and type(succs[0]) is GaussianLikelihood and succs[0].isotropic \
                    and preds[0].Nx <= preds[0].Nz and preds[0].rank == preds[0].Nx and succs[0]._y_dev is not None:
                self._rot[i] = dict(side="out", lin=preds[0], lik=succs[0], y_rot=None)

    # -- storage: one (a, b) pair per directed factor<->variable message
    def _allocate(self):
        ctx, B, L = self.ctx, self.batch, len(self.variables)
        self._snap = None
        g = self.model.dag
        self.adj = {v: list(g.predecessors(v)) + list(g.successors(v)) for v in self.variables}
        keys = []
        for v in self.variables:
            for f in self.adj[v]:
                keys += [(f, v), (v, f)]
        self.msg_keys = keys
        self.A = ctx.zeros((len(keys), B), np.float64)
        self.S = ctx.zeros((3 * L, B), np.float64)
        self.msg = {}
        # A SISO variable (one factor before, one after) passes on to one what it received from the other
        # (base.py:183-205), and each direction's messages change only in that direction's sweep, before the variable is
        # visited: its two outgoing messages ARE the incoming ones of the other side - the same buffers, no copy (the
        # chain scheduler stores its variables this way too).  NOT so for a leaf with two parents or a root with two
        # children: there an outgoing message of one direction would alias an incoming one of the OTHER direction, which
        # changes a sweep earlier than the copy would (measured: a different trajectory).  TRAMP_TREE_ALIAS=0: copies.
        import os
        alias = os.environ.get("TRAMP_TREE_ALIAS", "1") != "0"
        siso = {v for v in self.variables if len(list(g.predecessors(v))) == 1 and len(list(g.successors(v))) == 1}
        self._aliased = set()
        for i, (src, dst) in enumerate(keys):
            if alias and isinstance(src, Variable) and src in siso:
                continue                                       # filled in below
            var = src if isinstance(src, Variable) else dst
            n = int(np.prod(self.shapes[var.id]))
            b = ctx.zeros((B, n), self.dtype)
            b.version = 0
            self.msg[(src, dst)] = _Msg(self.A.row(i), b)
        if alias:
            for v in self.variables:
                if v in siso:
                    f0, f1 = self.adj[v]
                    self.msg[(v, f0)] = self.msg[(f1, v)]
                    self.msg[(v, f1)] = self.msg[(f0, v)]
                    self._aliased.update([(v, f0), (v, f1)])
            self.msg = {k: self.msg[k] for k in keys}          # keep the order of msg_keys (edge numbering of the monitor)
        self.r = [ctx.zeros((B, int(np.prod(self.shapes[v.id]))), self.dtype) for v in self.variables]
        self._const_msg = {}
        self._const_done = set()
        self._allocate_monitor(len(keys))
        self._allocated = True

    def init_message_dag(self, initializer):
        "message_passing.py:211-232: every message of both directions starts at the initializer's (a, b)"
        if not self._allocated:
            self._allocate()
        g = self.model.dag
        for (src, dst), m in self.msg.items():
            var = src if isinstance(src, Variable) else dst
            direction = "fwd" if g.has_edge(src, dst) else "bwd"
            shape = self.shapes[var.id]
            m.a.fill(initializer.init_a(shape, var.id, direction))
            value = initializer.init_b(shape, var.id, direction)
            fill_message(m.b, value)
            i = self.variables.index(var)
            if i in self._rot and np.any(np.asarray(value) != 0):         # initial messages live in the rotated basis
                m.b.copy_from(self._rotate(i, m.b, back=False))
            m.b.version += 1
        for r in self.r:
            r.fill(0.0)
        self._const_done = set()

    # -- alternating message sets
    def _latest(self, key):
        m = self.msg[key]
        if not self._pp_active() or id(m) in self._written:
            return m
        return self._msg_prev[key]

    def _old(self, key):
        return self._msg_prev[key] if self._pp_active() else self.msg[key]

    def _mark(self, m):
        m.b.version += 1
        if self._pp_active():
            self._written.add(id(m))

    def _allocate_snapshot(self):
        ctx, B = self.ctx, self.batch
        A2 = ctx.zeros(tuple(self.A.shape), np.float64)
        prev = {}
        for i, k in enumerate(self.msg_keys):
            if k in self._aliased:
                continue
            b = ctx.zeros(tuple(self.msg[k].b.shape), self.dtype)
            b.version = 0
            prev[k] = _Msg(A2.row(i), b)
        for v in self.variables:                               # the same aliases as in the live set
            f = self.adj[v]
            if len(f) == 2 and (v, f[0]) in self._aliased:
                prev[(v, f[0])] = prev[(f[1], v)]
                prev[(v, f[1])] = prev[(f[0], v)]
        self._msg_prev = {k: prev[k] for k in self.msg_keys}
        self._written = set()
        self._snap = dict(frozen={k: ctx.zeros(tuple(m.b.shape), self.dtype) for k, m in self.msg.items()}, A=A2)

    def _message_triples(self):
        s = self._snap
        return [(m.b, self._msg_prev[k].b, s["frozen"][k]) for k, m in self.msg.items()]

    def _sync_snapshot(self):
        for live, prev, _ in self._message_triples():
            prev.copy_from(live)
            prev.version += 1
        self._snap["A"].copy_from(self.A)
        self._written = set()

    def _swap_sets(self):
        self.msg, self._msg_prev = self._msg_prev, self.msg
        self.A, self._snap["A"] = self._snap["A"], self.A
        self._written = set()

    # -- node updates
    def _factor_send(self, f, direction):
        g = self.model.dag
        if f.n_prev > 1:
            return self._multi_input_send(f, direction)
        if f.n_next > 1:
            return self._multi_output_send(f, direction)
        prev = next(iter(g.predecessors(f)), None)
        nxt = next(iter(g.successors(f)), None)
        target = nxt if direction == "fwd" else prev
        if target is None:
            return
        out = self.msg[(f, target)]
        old = self._old((f, target))
        d = self.damping
        const = isinstance(f, (GaussianPrior, GaussianLikelihood))
        inp = self._latest((target, f))
        if const and not d and (f, direction, out.b.ptr) in self._const_done:
            if self._pp_active():
                self._written.add(id(out))                     # already in this buffer
            return
        if const and d:
            # constant message under damping: computed once, then msg <- d msg + (1 - d) c (as on the chain)
            if (f, direction) not in self._const_msg:
                ca = self.ctx.empty((self.batch,), np.float64)
                cb = self.ctx.empty(tuple(out.b.shape), self.dtype)
                if f.n_prev == 0:
                    f.compute_forward_message(inp.a, inp.b, out_a=ca, out_b=cb)
                else:
                    f.compute_backward_message(inp.a, inp.b, out_a=ca, out_b=cb)
                self._const_msg[(f, direction)] = (ca, cb)
            ca, cb = self._const_msg[(f, direction)]
            if old is not out:
                self.ctx._call("tramp_damp_to", dtype_code(self.dtype), out.b.shape[1], self.batch, 1.0 - float(d),
                               ca.ptr, cb.ptr, old.a.ptr, old.b.ptr, out.a.ptr, out.b.ptr)
            else:
                self.ctx._call("tramp_damp", dtype_code(self.dtype), out.b.shape[1], self.batch, 1.0 - float(d),
                               ca.ptr, cb.ptr, out.a.ptr, out.b.ptr)
            self._mark(out)
            return
        kw = dict(out_a=out.a, out_b=out.b)
        if d:
            kw.update(damping=d, old_a=old.a, old_b=old.b)
        if f.n_prev == 0:
            f.compute_forward_message(inp.a, inp.b, **kw)
        elif f.n_next == 0:
            f.compute_backward_message(inp.a, inp.b, **kw)
        else:
            z, x = self._latest((prev, f)), self._latest((nxt, f))
            if direction == "fwd":
                f.compute_forward_message(z.a, z.b, x.a, x.b, **kw)
            else:
                f.compute_backward_message(z.a, z.b, x.a, x.b, **kw)
        self._mark(out)
        if const and not d:
            self._const_done.add((f, direction, out.b.ptr))

    def _multi_input_send(self, f, direction):
        "K-input, one-output factor (SumChannel): forward to its output, backward one message per input"
        g = self.model.dag
        prevs = list(g.predecessors(f))
        nxt = next(iter(g.successors(f)))
        zs = [self._latest((p, f)) for p in prevs]
        x = self._latest((nxt, f))
        az, bz = [m.a for m in zs], [m.b for m in zs]
        d = self.damping
        targets = [(nxt, None)] if direction == "fwd" else [(p, k) for k, p in enumerate(prevs)]
        for tgt, k in targets:
            out, old = self.msg[(f, tgt)], self._old((f, tgt))
            kw = dict(out_a=out.a, out_b=out.b)
            if d:
                kw.update(damping=d, old_a=old.a, old_b=old.b)
            if k is None:
                f.compute_forward_message(az, bz, x.a, x.b, **kw)
            else:
                f.compute_backward_message(az, bz, x.a, x.b, target=k, **kw)
            self._mark(out)

    def _multi_output_send(self, f, direction):
        "one-input, K-output factor (DuplicateChannel): forward one message per output, backward to its input"
        g = self.model.dag
        prev = next(iter(g.predecessors(f)))
        nxts = list(g.successors(f))
        z = self._latest((prev, f))
        xs = [self._latest((x, f)) for x in nxts]
        ax, bx = [m.a for m in xs], [m.b for m in xs]
        d = self.damping
        targets = [(prev, None)] if direction == "bwd" else [(x, k) for k, x in enumerate(nxts)]
        for tgt, k in targets:
            out, old = self.msg[(f, tgt)], self._old((f, tgt))
            kw = dict(out_a=out.a, out_b=out.b)
            if d:
                kw.update(damping=d, old_a=old.a, old_b=old.b)
            if k is None:
                f.compute_backward_message(z.a, z.b, ax, bx, **kw)
            else:
                f.compute_forward_message(z.a, z.b, ax, bx, target=k, **kw)
            self._mark(out)

    def _variable_send(self, v, targets):
        "Variable.forward_message / backward_message (tramp/base.py:186-211)"
        nbrs = self.adj[v]
        inc = [self._latest((f, v)) for f in nbrs]
        for f in targets:
            if (v, f) in self._aliased:
                continue                                       # the outgoing message is the other side's incoming buffer
            out = self.msg[(v, f)]
            if len(nbrs) == 2:
                other = inc[1 - nbrs.index(f)]
                out.a.copy_from(other.a)
                out.b.copy_from(other.b)
            else:
                k = len(nbrs)
                ap = (C.c_void_p * k)(*[m.a.ptr for m in inc])
                bp = (C.c_void_p * k)(*[m.b.ptr for m in inc])
                self.ctx._call("tramp_message_sum", dtype_code(self.dtype), out.b.shape[1], self.batch, k, ap, bp,
                               nbrs.index(f), out.a.ptr, out.b.ptr)
            self._mark(out)

    def forward_message(self):
        "message_passing.py:249-255"
        g = self.model.dag
        for node in self.model.forward_ordering:
            if isinstance(node, Factor):
                self._factor_send(node, "fwd")
            else:
                self._variable_send(node, list(g.successors(node)))

    def backward_message(self):
        "message_passing.py:257-263"
        g = self.model.dag
        for node in reversed(self.model.forward_ordering):
            if isinstance(node, Factor):
                self._factor_send(node, "bwd")
            else:
                self._variable_send(node, list(g.predecessors(node)))

    def update_variables(self):
        """message_passing.py:265-269: after the backward sweep the message a variable sent to its first
        parent is the sum of everything else, so posterior = (parent -> v) + (v -> parent)."""
        dt = dtype_code(self.dtype)
        for i, v in enumerate(self.variables):
            f = self.adj[v][0]
            fv, vf = self.msg[(f, v)], self.msg[(v, f)]
            n = self.r[i].shape[1]
            if self._device_monitor:
                # previous mean read from r, new one written to r_prev, then the two swap roles: the iterate before
                # an increase stays at hand for an instance that gives up (as the chain scheduler)
                self.ctx._call("tramp_variable_posterior", dt, n, self.batch, fv.a.ptr, fv.b.ptr, vf.a.ptr, vf.b.ptr,
                               self.r[i].ptr, self.r_prev[i].ptr, self.S.row(3 * i).ptr, self.S.row(3 * i + 1).ptr,
                               self.S.row(3 * i + 2).ptr)
                self.r[i], self.r_prev[i] = self.r_prev[i], self.r[i]
            else:
                self.ctx._call("tramp_variable_posterior", dt, n, self.batch, fv.a.ptr, fv.b.ptr, vf.a.ptr, vf.b.ptr,
                               self.r[i].ptr, self.r[i].ptr, self.S.row(3 * i).ptr, self.S.row(3 * i + 1).ptr,
                               self.S.row(3 * i + 2).ptr)

    def _edge_name(self, e):
        src, dst = self.msg_keys[e]
        return f"message {getattr(src, 'id', src)} -> {getattr(dst, 'id', dst)}"

    def check_message(self):
        bad = np.isnan(self.A_host)
        if bad.any():
            e, b = np.argwhere(bad)[0]
            raise ValueError(self._edge_name(int(e)) + f" (instance {b}): a is nan")

    def get_edges_data(self):
        """a of every factor->variable message as {(variable id, direction): array[batch]}, like the chain
        scheduler; where a variable has several factors on one side the direction carries the factor id,
        e.g. ("z", "fwd[f_3]")."""
        g = self.model.dag
        out = {}
        A = self.A_host.copy()
        done = self.status != 0
        if done.any() and self._device_monitor:
            A[:, done] = self._frozen_A[:, done]              # a stopped instance: the values of its stopping iteration
        for i, (src, dst) in enumerate(self.msg_keys):
            if not isinstance(dst, Variable):
                continue
            fwd = g.has_edge(src, dst)
            several = (dst.n_prev if fwd else dst.n_next) > 1
            direction = ("fwd" if fwd else "bwd") + (f"[{src.id}]" if several else "")
            out[(dst.id, direction)] = A[i].copy()
        return out

    def get_message(self, source, target):
        """(a, b) of the message source -> target; nodes are given as objects or ids."""
        def find(x):
            for n in self.model.forward_ordering:
                if n is x or getattr(n, "id", None) == x:
                    return n
            raise KeyError(x)
        src, dst = find(source), find(target)
        m = self.msg[(src, dst)]
        var = src if isinstance(src, Variable) else dst
        i = self.variables.index(var)
        a, b = m.a.get(), (self._rotate(i, m.b, back=True) if i in self._rot else m.b).get()
        return (float(a[0]), b[0]) if self.batch == 1 else (a, b)

    def log_evidence(self):
        raise NotImplementedError("log_evidence is implemented for chain models")


class TreeStateEvolution:
    """Scalar state evolution on a tree-structured model (tramp/algos/state_evolution.py:5-27 with
    Variable.forward_state_evolution / backward_state_evolution, tramp/base.py:213-233): the sweeps of
    `TreeExpectationPropagation` on the precisions ``a`` alone; factor errors are the closed forms or the
    device quadratures K13 / K14 (factor.compute_*_state_evolution)."""

    def __init__(self, model):
        from .algos import EarlyStopping
        self.model = model
        model.init_second_moments()
        self.variables = model.variables
        self.taus = model.get_second_moments()
        self.n_iter = 0
        self.default_stopping = EarlyStopping()

    def _factor_send(self, f, direction, d):
        g = self.model.dag
        if f.n_prev > 1:                                    # K-input factor (SumChannel)
            prevs = list(g.predecessors(f))
            nxt = next(iter(g.successors(f)))
            az, taus = [self.msg[(p, f)] for p in prevs], [self.taus[p.id] for p in prevs]
            if direction == "fwd":
                news, tgts = [f.compute_forward_state_evolution(az, self.msg[(nxt, f)], taus)], [nxt]
            else:
                news, tgts = f.compute_backward_state_evolution(az, self.msg[(nxt, f)], taus), prevs
            for new, t in zip(news, tgts):
                if np.any(np.isnan(new)):
                    raise ValueError(f"{f.id}->{t.id} a is nan")
                self.msg[(f, t)] = d * self.msg[(f, t)] + (1 - d) * new if d else new
            return
        if f.n_next > 1:                                    # K-output factor (DuplicateChannel)
            prev = next(iter(g.predecessors(f)))
            nxts = list(g.successors(f))
            ax = [self.msg[(x, f)] for x in nxts]
            if direction == "fwd":
                news, tgts = f.compute_forward_state_evolution(self.msg[(prev, f)], ax, self.taus[prev.id]), nxts
            else:
                news, tgts = [f.compute_backward_state_evolution(self.msg[(prev, f)], ax, self.taus[prev.id])], [prev]
            for new, t in zip(news, tgts):
                if np.any(np.isnan(new)):
                    raise ValueError(f"{f.id}->{t.id} a is nan")
                self.msg[(f, t)] = d * self.msg[(f, t)] + (1 - d) * new if d else new
            return
        prev = next(iter(g.predecessors(f)), None)
        nxt = next(iter(g.successors(f)), None)
        target = nxt if direction == "fwd" else prev
        if target is None:
            return
        if f.n_prev == 0:
            new = f.compute_forward_state_evolution(self.msg[(nxt, f)])
        elif f.n_next == 0:
            new = f.compute_backward_state_evolution(self.msg[(prev, f)], self.taus[prev.id])
        elif direction == "fwd":
            new = f.compute_forward_state_evolution(self.msg[(prev, f)], self.msg[(nxt, f)], self.taus[prev.id])
        else:
            new = f.compute_backward_state_evolution(self.msg[(prev, f)], self.msg[(nxt, f)], self.taus[prev.id])
        if np.any(np.isnan(new)):
            raise ValueError(f"{f.id}->{target.id} a is nan")
        self.msg[(f, target)] = d * self.msg[(f, target)] + (1 - d) * new if d else new

    def _variable_send(self, v, targets):
        a_hat = sum(self.msg[(f, v)] for f in self.adj[v])
        for f in targets:
            self.msg[(v, f)] = a_hat - self.msg[(f, v)]

    def iterate(self, max_iter=200, callback=None, initializer=None, damping=None, warm_start=False, update_dA=False):
        from .algos import ConstantInit
        if update_dA:
            raise NotImplementedError("update_dA is not tracked")
        initializer = initializer or ConstantInit(a=0, b=0)
        callback = callback or self.default_stopping
        d = float(damping) if damping else 0.0
        g = self.model.dag
        if not warm_start:
            self.adj = {v: list(g.predecessors(v)) + list(g.successors(v)) for v in self.variables}
            self.msg = {}
            for v in self.variables:
                for f in self.adj[v]:
                    for src, dst in ((f, v), (v, f)):
                        self.msg[(src, dst)] = initializer.init_a(None, v.id, "fwd" if g.has_edge(src, dst) else "bwd")
            self.n_iter = 0
        order = self.model.forward_ordering
        for it in range(max_iter):
            for node in order:
                if isinstance(node, Factor):
                    self._factor_send(node, "fwd", d)
                else:
                    self._variable_send(node, list(g.successors(node)))
            for node in reversed(order):
                if isinstance(node, Factor):
                    self._factor_send(node, "bwd", d)
                else:
                    self._variable_send(node, list(g.predecessors(node)))
            self.v = [1. / sum(self.msg[(f, v)] for f in self.adj[v]) for v in self.variables]     # base.py:167-170
            self.n_iter += 1
            if callback(self, it, max_iter):
                break
        return self

    def get_variables_data(self, ids="all", want_r=False):
        return {v.id: dict(v=self.v[i]) for i, v in enumerate(self.variables) if ids == "all" or v.id in ids}

    def get_variable_data(self, id):
        return self.get_variables_data([id])[id]

    def get_edges_data(self):
        return {(getattr(s, "id", s), getattr(t, "id", t)): a for (s, t), a in self.msg.items()}
