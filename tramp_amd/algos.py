"""Message-passing schedulers (host Python, CPU) driving the HIP factor kernels.

Counterpart of tramp/algos/{message_passing,expectation_propagation,state_evolution,
callbacks,initial_conditions}.py for SISO chains:

  iterate():  forward sweep in topological order, backward sweep in reverse order,
              variable posteriors, callback   (message_passing.py:330-357)
  init:       ConstantInit(a=0, b=0)           (initial_conditions.py:13-24)
  damping:    constant, on factor->variable edges, both a and b (message_passing.py:70-127)
  NaN guard:  ValueError when a message precision is NaN (message_passing.py:187-209)
  stopping:   EarlyStoppingEP on ||r_new - r_old|| / ||r_new|| (callbacks.py:250-286),
              EarlyStopping on |v_new - v_old| for SE (callbacks.py:195-243)

All messages stay resident in HBM: per variable two (a[batch], b[batch, n]) pairs,
updated in place by the factor kernels (damping fused).  One device->host read of
the scalar table per iteration feeds the NaN guard and the stopping rule.
"""
import logging

import numpy as np

from . import _lib
from .base import Factor, Variable
from .device import default_context, dtype_code
from .priors import GaussianPrior
from .likelihoods import GaussianLikelihood

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------ initial conditions
class ConstantInit:
    "tramp/algos/initial_conditions.py:13-24"

    def __init__(self, a=0, b=0):
        self.a = a
        self.b = b

    def init_a(self, shape, id, direction):
        return self.a

    def init_b(self, shape, id, direction):
        return self.b


class CustomInit(ConstantInit):
    "Per-(variable id, direction) values (initial_conditions.py:45-85): b a constant or an array of the variable's shape."

    def __init__(self, a_init=None, b_init=None, a=0, b=0):
        super().__init__(a, b)
        self.a_init = {(i, d): v for i, d, v in (a_init or [])}
        self.b_init = {(i, d): v for i, d, v in (b_init or [])}

    def init_a(self, shape, id, direction):
        return self.a_init.get((id, direction), self.a)

    def init_b(self, shape, id, direction):
        return self.b_init.get((id, direction), self.b)


class NoisyInit(ConstantInit):
    "Random initial messages (tramp/algos/initial_conditions.py:27-42); every instance of an ensemble draws its own b"

    def __init__(self, a_mean=0, a_var=0, b_mean=0, b_var=1):
        self.a_mean, self.a_var, self.b_mean, self.b_var = a_mean, a_var, b_mean, b_var
        self.a_sigma, self.b_sigma = np.sqrt(a_var), np.sqrt(b_var)

    def init_a(self, shape, id, direction):
        return self.a_mean + self.a_sigma * np.random.standard_normal()

    def init_b(self, shape, id, direction):
        return self.b_mean + self.b_sigma * np.random.standard_normal(shape)


def fill_message(buf, value):
    "initial b of one message: a constant, or an array of the variable's shape (shared by the instances) / [batch, n]"
    if np.ndim(value) == 0:
        buf.fill(float(value))
        return
    h = np.asarray(value, dtype=buf.dtype)
    B, n = buf.shape
    h = h.reshape(B, n) if h.size == B * n and B > 1 else np.broadcast_to(h.reshape(1, n), (B, n))
    buf.set(np.ascontiguousarray(h))


# ------------------------------------------------------------------ callbacks
class PassCallback:
    def __call__(self, algo, i, max_iter):
        return False


class JoinCallback:
    "tramp/algos/callbacks.py:22-32"

    def __init__(self, callbacks):
        self.callbacks = callbacks

    def __call__(self, algo, i, max_iter):
        return any([cb(algo, i, max_iter) for cb in self.callbacks])


class EarlyStoppingEP:
    """tramp/algos/callbacks.py:250-286; with an ensemble every instance must be
    below ``tol`` (``algo.n_iter_instance`` records when each one first was)."""

    def __init__(self, ids="all", tol=1e-6, wait_increase=5, max_increase=0.2):
        self.ids = ids
        self.tol = tol
        self.wait_increase = wait_increase
        self.max_increase = max_increase

    def __call__(self, algo, i, max_iter):
        # host-side evaluation (the chain scheduler runs this rule on the device instead when it covers all
        # variables, csrc/monitor.hip); per instance: an instance that meets the rule, or gives up after an
        # increase, is frozen and the others go on
        if i == 0:
            return False                      # old_rs = None on the first iteration
        tols = algo.relative_change(self.ids)          # [n_vars, batch]
        with np.errstate(invalid="ignore"):
            worst = tols.max(axis=0)
            algo.mark_converged(worst < self.tol)
            if i > self.wait_increase:
                algo.mark_diverged(worst > self.max_increase)
        if np.all(algo.status != 0):
            logger.info("early stopping: every instance is below tol=%.2e or has given up", self.tol)
            return True
        return False


class EarlyStopping:
    "tramp/algos/callbacks.py:195-243 (state evolution: change of v)"

    def __init__(self, ids="all", tol=1e-6, min_variance=-1, wait_increase=5, max_increase=0.2):
        self.ids, self.tol, self.min_variance = ids, tol, min_variance
        self.wait_increase, self.max_increase = wait_increase, max_increase
        self.old_vs = None

    def __call__(self, algo, i, max_iter):
        if i == 0:
            self.old_vs = None
        data = algo.get_variables_data(self.ids)
        new_vs = [np.asarray(d["v"], dtype=float) for d in data.values()]
        if any(np.any(v < self.min_variance) for v in new_vs):
            return True
        if any(np.any(np.isnan(v)) for v in new_vs):
            logger.warning("early stopping nan values")
            return True
        if self.old_vs is not None:
            tols = [np.abs(o - n) for o, n in zip(self.old_vs, new_vs)]
            if max(np.max(t) for t in tols) < self.tol:
                return True
            inc = [n - o for o, n in zip(self.old_vs, new_vs)]
            if i > self.wait_increase and max(np.max(t) for t in inc) > self.max_increase:
                return True
        self.old_vs = new_vs
        return False


class TrackEvolution:
    "tramp/algos/callbacks.py:88-108"

    def __init__(self, ids="all", every=1):
        self.ids, self.every, self.records = ids, every, []

    def __call__(self, algo, i, max_iter):
        if i == 0:
            self.records = []
        if i % self.every == 0:
            for vid, d in algo.get_variables_data(self.ids, want_r=False).items():
                self.records.append(dict(id=vid, v=d["v"], iter=i))
        return False

    def get_dataframe(self):
        "the records as a pandas DataFrame (tramp/algos/callbacks.py: get_dataframe of every tracking callback)"
        import pandas as pd
        return pd.DataFrame(self.records)


class TrackErrors:
    """tramp/algos/callbacks.py:131-162: the metrics of tramp/algos/metrics.py against the true signal, per instance.  The default
    ["mse"] is reduced on the device; "sign_mse", "phase_mse", "overlap" are scored on the host from the posterior means."""

    def __init__(self, true_values, metrics=("mse",), every=1):
        from .metrics import METRICS
        self.true_values, self.metrics, self.every, self.records = true_values, list(metrics), every, []
        for m in self.metrics:
            if m not in METRICS:
                raise KeyError(m)

    def __call__(self, algo, i, max_iter):
        from .metrics import METRICS
        if i == 0:
            self.records = []
        if i % self.every == 0:
            host = [m for m in self.metrics if m != "mse"]
            data = algo.get_variables_data(list(self.true_values)) if host else {}
            for vid, x in self.true_values.items():
                rec = dict(id=vid, iter=i)
                if "mse" in self.metrics:
                    rec["mse"] = algo.mse(vid, x)
                for m in host:
                    rec[m] = METRICS[m](x, data[vid]["r"], batch=getattr(algo, "batch", None))
                self.records.append(rec)
        return False

    def get_dataframe(self):
        "the records as a pandas DataFrame (tramp/algos/callbacks.py: get_dataframe of every tracking callback)"
        import pandas as pd
        return pd.DataFrame(self.records)


class TrackEstimate:
    "tramp/algos/callbacks.py:111-128: the posterior means, every `every` iterations"

    def __init__(self, ids="all", every=1):
        self.ids, self.every, self.records = ids, every, []

    def __call__(self, algo, i, max_iter):
        if i == 0:
            self.records = []
        if i % self.every == 0:
            for vid, d in algo.get_variables_data(self.ids).items():
                self.records.append(dict(id=vid, r=d["r"], iter=i))
        return False

    def get_dataframe(self):
        "the records as a pandas DataFrame (tramp/algos/callbacks.py: get_dataframe of every tracking callback)"
        import pandas as pd
        return pd.DataFrame(self.records)


class TrackMessages:
    "tramp/algos/callbacks.py:49-62: the scalar part a of every factor->variable message"

    def __init__(self, keys=("a", "n_iter", "direction")):
        self.keys, self.records = tuple(keys), []

    def __call__(self, algo, i, max_iter):
        if i == 0:
            self.records = []
        for (vid, direction), a in algo.get_edges_data().items():
            rec = dict(a=float(a[0]) if a.size == 1 else a, n_iter=algo.n_iter, direction=direction, id=vid)
            self.records.append({k: rec[k] for k in (*self.keys, "id") if k in rec})
        return False

    def get_dataframe(self):
        "the records as a pandas DataFrame (tramp/algos/callbacks.py: get_dataframe of every tracking callback)"
        import pandas as pd
        return pd.DataFrame(self.records)


class TrackObjective:
    "tramp/algos/callbacks.py:65-85, model level: the EP log-evidence A_model per iteration"

    def __init__(self):
        self.model_records = []

    def __call__(self, algo, i, max_iter):
        if i == 0:
            self.model_records = []
        self.model_records.append(dict(A=algo.log_evidence(), n_iter=algo.n_iter))
        return False


class TrackOverlaps:
    "tramp/algos/callbacks.py:165-191: m = r.x/N, q = r.r/N, Q = x.x/N against the true signal"

    def __init__(self, true_values, ids="all", every=1):
        self.true_values, self.ids, self.every, self.records = true_values, ids, every, []

    def __call__(self, algo, i, max_iter):
        if i == 0:
            self.records = []
        if i % self.every == 0:
            for vid, d in algo.get_variables_data(self.ids).items():
                if vid not in self.true_values:
                    continue
                x = np.asarray(self.true_values[vid], dtype=np.float64)
                r = np.asarray(d["r"], dtype=np.float64).reshape(x.shape)
                n = x.shape[-1]
                self.records.append(dict(id=vid, m=np.sum(r * x, axis=-1) / n, q=np.sum(r * r, axis=-1) / n,
                                         Q=np.sum(x * x, axis=-1) / n, iter=i))
        return False

    def get_dataframe(self):
        "the records as a pandas DataFrame (tramp/algos/callbacks.py: get_dataframe of every tracking callback)"
        import pandas as pd
        return pd.DataFrame(self.records)


class LogProgress:
    "tramp/algos/callbacks.py:33-46"

    def __init__(self, ids="all", every=1):
        self.ids, self.every = ids, every

    def __call__(self, algo, i, max_iter):
        if i % self.every == 0:
            import logging
            log = logging.getLogger(__name__)
            log.info(f"iteration={i+1}/{max_iter}")
            for vid, d in algo.get_variables_data(self.ids, want_r=False).items():
                log.info(f"id={vid} v={np.mean(d['v']):.3f}")
        return False


# ------------------------------------------------------------------ EP
class ExpectationPropagation:
    """Gaussian EP on a chain model (tramp/algos/expectation_propagation.py:5-32).

    ``batch`` independent instances share the model structure (and the linear
    channels' matrices); it is inferred from the observations' leading axis.
    """

    def __new__(cls, model, *args, **kwargs):
        # tree-structured models (a variable with several parents or children) run on the general scheduler
        if cls is ExpectationPropagation and not getattr(model, "is_chain", True):
            from .tree import TreeExpectationPropagation
            return super().__new__(TreeExpectationPropagation)
        return super().__new__(cls)

    def __init__(self, model, batch=None, ctx=None):
        self.model = model
        self.shapes = model.init_shapes()
        self.variables = model.variables
        self.factors = model.factors
        for v in self.variables:
            if v.n_next != 1 or v.n_prev != 1:
                raise ValueError(f"variable {v.id}: EP runs on observed models (every variable between "
                                 "two factors); call model.to_observed(...) first")
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
        self.default_stopping = EarlyStoppingEP()
        self._find_rotations()

    # -- rotated output basis ------------------------------------------------------------
    def _find_rotations(self):
        """A dense LinearChannel (M <= N) whose output variable is observed through an isotropic
        GaussianLikelihood: that likelihood is rotation invariant, so the variable is carried in the
        channel's left singular basis, z' = Uk^T z, with y' = Uk^T y (rotated once).  Messages, norms,
        variances and every other variable are unchanged; the two passes over Uk leave the iteration
        (tramp_linear_set_rotated).  Whatever is read of that variable (posterior mean, messages) is
        rotated back on the way out, so callers and callbacks see the reference's quantities.
        TRAMP_ROTATE_OUTPUT=0 / TRAMP_ROTATE_INPUT=0 disable the two sides."""
        import os
        self._rot = {}
        from .channels import LinearChannel
        for i, v in enumerate(self.variables):
            if os.environ.get("TRAMP_ROTATE_OUTPUT", "1") == "0":
                break
            lin, lik = self.model.prev_factor(v), self.model.next_factor(v)
            if type(lin) is LinearChannel and type(lik) is GaussianLikelihood and lik.isotropic \
                    and lin.Nx <= lin.Nz and lin.rank == lin.Nx and lik._y_dev is not None:
                self._rot[i] = dict(side="out", lin=lin, lik=lik, y_rot=None)
        # mirror image on the input side: an isotropic zero-mean GaussianPrior feeding a dense LinearChannel.
        # The variable is carried as z' = Vt z (k numbers per instance); the part of z orthogonal to the rows
        # of Vt only ever sees the prior's b = 0 and stays exactly zero, so nothing is lost (zero initial
        # messages required - checked in init_message_dag).  Removes the two passes over Vt.
        if os.environ.get("TRAMP_ROTATE_INPUT", "1") != "0":
            for i, v in enumerate(self.variables):
                pri, lin = self.model.prev_factor(v), self.model.next_factor(v)
                if type(pri) is GaussianPrior and type(lin) is LinearChannel and pri.isotropic \
                        and np.all(np.asarray(pri.mean) == 0) and lin.rank == min(lin.Nx, lin.Nz):
                    self._rot[i] = dict(side="in", lin=lin, k=min(lin.Nx, lin.Nz))

    def _auto_attach_digits(self):
        """TRAMP_MATVEC_DIGITS=auto (the default): an ensemble of >= 3 instances runs its dense passes on the int8-digit
        engine (csrc/gemm_sliced.h: 7 digits for f64 factors, 4 for f32 - at least the accuracy of the matrix-core pass) if
        the copies fit beside what is already allocated (2 forms x digits bytes per matrix element of Vt, and of Uk unless
        the output side is carried in Uk's basis) with 8 GiB to spare.  One attach per channel (~1 s at 50 000 x 100 000);
        LinearChannel(digits=0) or TRAMP_MATVEC_DIGITS=0 opts out, explicit digits are left alone."""
        if self.batch < 3:
            return
        from .channels import LinearChannel
        masks = self._lin_masks()
        for f in self.factors:
            if type(f) is not LinearChannel or f.digits or not getattr(f, "_digits_auto", False):
                continue
            uk = not (masks.get(f, 0) & 1)
            digits = 7 if self.dtype == np.float64 else 4
            need = 2 * digits * (f._Vt.shape[0] * f._Vt.shape[1] + (f._Uk.shape[0] * f._Uk.shape[1] if uk else 0))
            free, _ = self.ctx.mem_info()
            if free < need * 1.1 + (8 << 30):
                f._digits_auto = False                 # asked once: not on this device
                continue
            f.attach_digits(digits, uk=uk)
            f._digits_auto = False

    def _lin_masks(self):
        "LinearChannel -> tramp_linear_set_rotated mask (1 output side, 2 input side)"
        masks = {}
        for rot in self._rot.values():
            masks[rot["lin"]] = masks.get(rot["lin"], 0) | (1 if rot["side"] == "out" else 2)
        return masks

    def _apply_basis(self, on):
        for lin, mask in self._lin_masks().items():
            self.ctx._call("tramp_linear_set_rotated", lin._handle, mask if on else 0)
            lin._rot_mask = mask if on else 0
            lin._t_key = lin._u_key = None
        for i, rot in self._rot.items():
            if rot["side"] != "out":
                continue
            if on:
                if rot["y_rot"] is None:
                    rot["y_rot"] = self._rotate(i, rot["lik"]._y_dev, back=False)
                rot["y_saved"], rot["lik"]._y_dev = rot["lik"]._y_dev, rot["y_rot"]
            else:
                rot["lik"]._y_dev = rot.pop("y_saved", rot["lik"]._y_dev)

    def _rotate(self, i, src, back):
        """Output side: Uk^T src (into the rotated basis) or Uk src (back), [batch, M].  Input side: back only,
        Vt^T src, [batch, k] -> [batch, N]."""
        rot = self._rot[i]
        lin, dt = rot["lin"], dtype_code(self.dtype)
        if rot["side"] == "out":
            dst = self.ctx.empty(tuple(src.shape), self.dtype)
            self.ctx._call("tramp_matvec_nt" if back else "tramp_matvec_nn", dt, 0, lin.Nx, lin.Nx,
                           src.shape[0], lin._Uk.ptr, src.ptr, dst.ptr)
            return dst
        assert back, "the input side is only ever expanded"
        dst = self.ctx.empty((src.shape[0], lin.Nz), self.dtype)
        self.ctx._call("tramp_matvec_nn", dt, 0, rot["k"], lin.Nz, src.shape[0], lin._Vt.ptr, src.ptr, dst.ptr)
        return dst

    class _Basis:
        "while active: the rotated channels skip Uk and their likelihoods present y' = Uk^T y"

        def __init__(self, ep):
            self.ep = ep

        def __enter__(self):
            ep = self.ep
            ep._basis_depth = getattr(ep, "_basis_depth", 0) + 1
            if ep._basis_depth == 1:
                ep._apply_basis(True)
            return self

        def __exit__(self, *exc):
            ep = self.ep
            ep._basis_depth -= 1
            if ep._basis_depth == 0:
                ep._apply_basis(False)
            return False

    # -- storage
    def _allocate(self):
        ctx, B, L = self.ctx, self.batch, len(self.variables)
        self._snap = None                                    # snapshot / frozen messages belong to the old buffers
        self.A = ctx.zeros((2 * L, B), np.float64)           # a of every factor->variable edge
        self.S = ctx.zeros((3 * L, B), np.float64)           # per variable: v, sum (r-r_old)^2, sum r^2
        self.a_fwd = [self.A.row(2 * i) for i in range(L)]
        self.a_bwd = [self.A.row(2 * i + 1) for i in range(L)]
        self.b_fwd, self.b_bwd, self.r = [], [], []
        for i, v in enumerate(self.variables):
            n = int(np.prod(self.shapes[v.id]))
            if i in self._rot and self._rot[i]["side"] == "in":
                n = self._rot[i]["k"]                           # carried as z' = Vt z
            self.b_fwd.append(ctx.zeros((B, n), self.dtype))
            self.b_bwd.append(ctx.zeros((B, n), self.dtype))
            self.r.append(ctx.zeros((B, n), self.dtype))
        for buf in self.b_fwd + self.b_bwd:
            buf.version = 0
        self._const_msg = {}
        self._const_done = set()
        self._allocate_monitor(2 * L)
        self._allocated = True

    def _allocate_monitor(self, n_edges):
        """Device-side bookkeeping (csrc/monitor.hip): previous posterior means (the variable-posterior
        kernel ping-pongs between r and r_prev, so the previous iterate costs no traffic), what a stopped
        instance reports (frozen r, v, a), and the state words the host polls."""
        ctx, B, L = self.ctx, self.batch, len(self.variables)
        self.r_prev = [ctx.zeros(r.shape, self.dtype) for r in self.r]
        self._frozen_r = [ctx.zeros(r.shape, self.dtype) for r in self.r]
        self._state = ctx.zeros((4 + 2 * B,), np.int32)
        self._v_prev = ctx.zeros((L, B), np.float64)
        self._v_frozen = ctx.zeros((L, B), np.float64)
        self._a_frozen = ctx.zeros((n_edges, B), np.float64)
        self._n_edges = n_edges
        self._reset_monitor()

    def _reset_monitor(self):
        B, L = self.batch, len(self.variables)
        self.ctx._call("tramp_memset", self._state, 0, self._state.nbytes)
        self.status = np.zeros(B, dtype=int)                 # 0 running / max_iter, 1 converged, 2 diverged
        self.n_iter_instance = np.zeros(B, dtype=int)
        self._frozen_v = np.zeros((L, B))
        self._frozen_A = np.zeros((self._n_edges, B))

    def init_message_dag(self, initializer):
        "message_passing.py:211-232"
        for i in [j for j, r in self._rot.items() if r["side"] == "in"]:
            v = self.variables[i]
            if any(np.any(np.asarray(initializer.init_b(self.shapes[v.id], v.id, d)) != 0) for d in ("fwd", "bwd")):
                del self._rot[i]                                # non-zero start: keep this variable in full
                self._allocated = False
        if not self._allocated:
            self._allocate()
        for i, v in enumerate(self.variables):
            shape = self.shapes[v.id]
            if i in self._rot and self._rot[i]["side"] == "in":
                self.a_fwd[i].fill(initializer.init_a(shape, v.id, "fwd"))
                self.a_bwd[i].fill(initializer.init_a(shape, v.id, "bwd"))
                for buf in (self.b_fwd[i], self.b_bwd[i], self.r[i]):
                    buf.fill(0.0)
                self.b_fwd[i].version += 1
                self.b_bwd[i].version += 1
                continue
            self.a_fwd[i].fill(initializer.init_a(shape, v.id, "fwd"))
            self.a_bwd[i].fill(initializer.init_a(shape, v.id, "bwd"))
            fill_message(self.b_fwd[i], initializer.init_b(shape, v.id, "fwd"))
            fill_message(self.b_bwd[i], initializer.init_b(shape, v.id, "bwd"))
            self.r[i].fill(0.0)
            if i in self._rot and self._rot[i]["side"] == "out":  # initial messages live in the rotated basis too
                for buf, d in ((self.b_fwd[i], "fwd"), (self.b_bwd[i], "bwd")):
                    if np.any(np.asarray(initializer.init_b(shape, v.id, d)) != 0):
                        buf.copy_from(self._rotate(i, buf, back=False))
            self.b_fwd[i].version += 1
            self.b_bwd[i].version += 1
        self._const_done = set()
        self._const_msg = {}
        self.ctx._call("tramp_prior_hint", 0, 0.0, 0.0, 0.0, 0, 0, None, None)   # messages rewritten: no stale moments

    # -- one factor update, in place, damping fused where the kernel supports it
    def _update(self, factor, direction, i_in, i_out):
        d = self.damping
        out_a = self.a_fwd[i_out] if direction == "fwd" else self.a_bwd[i_out]
        out_b = self.b_fwd[i_out] if direction == "fwd" else self.b_bwd[i_out]
        # Alternating message buffers (restore_messages on the chain scheduler, _take_snapshot): the set written in
        # the previous iteration IS the snapshot.  This iteration's messages go into the other set; the previous value
        # of a message (for damping) and the backward messages the forward sweep reads - not rewritten yet in this
        # iteration - come from the previous set.
        pp = self._pp_active()
        prev = self._snap if pp else None
        old_a = (prev["a_fwd"] if direction == "fwd" else prev["a_bwd"])[i_out] if pp else out_a
        old_b = (prev["fwd"] if direction == "fwd" else prev["bwd"])[i_out] if pp else out_b
        a_bwd, b_bwd = (prev["a_bwd"], prev["bwd"]) if (pp and direction == "fwd") else (self.a_bwd, self.b_bwd)
        const = isinstance(factor, (GaussianPrior, GaussianLikelihood))
        if const and not d and (factor, direction, out_b.ptr) in self._const_done:
            return          # undamped constant message: already in place (in this buffer), bit-identical every iteration
        if const and d:
            # Constant factor under damping: msg <- d msg + (1 - d) c with the same (c_a, c_b) every
            # iteration (K12).  c is computed once; the update is ONE in-place kernel - tramp_damp with
            # the roles swapped, (1 - d) c + d msg, the same products and sum, so the same bits.
            key = (i_out, direction)
            if key not in self._const_msg:
                ca = self.ctx.empty((self.batch,), np.float64)
                cb = self.ctx.empty(tuple(out_b.shape), self.dtype)
                if factor.n_prev == 0:
                    factor.compute_forward_message(a_bwd[i_out], b_bwd[i_out], out_a=ca, out_b=cb)
                else:
                    factor.compute_backward_message(self.a_fwd[i_out], self.b_fwd[i_out], out_a=ca, out_b=cb)
                self._const_msg[key] = (ca, cb)
            ca, cb = self._const_msg[key]
            old_key = (old_b.ptr, old_b.version, self.batch)
            if pp:          # (1 - d) c + d previous message, written into this iteration's buffer: the same products and sum
                self.ctx._call("tramp_damp_to", dtype_code(self.dtype), out_b.shape[1], self.batch, 1.0 - float(d),
                               ca.ptr, cb.ptr, old_a.ptr, old_b.ptr, out_a.ptr, out_b.ptr)
            else:
                self.ctx._call("tramp_damp", dtype_code(self.dtype), out_b.shape[1], self.batch, 1.0 - float(d),
                               ca.ptr, cb.ptr, out_a.ptr, out_b.ptr)
            out_b.version += 1
            # the consumer's cached projection of this message follows by linearity (O(k), no matrix pass)
            var = self.variables[i_out]
            consumer = self.model.prev_factor(var) if direction == "bwd" else self.model.next_factor(var)
            if hasattr(consumer, "affine_update"):
                consumer.affine_update("x" if direction == "bwd" else "z", old_key, out_b, cb, float(d),
                                       token=(id(factor), direction))
            return
        kw = dict(out_a=out_a, out_b=out_b)
        if d:
            kw.update(damping=d, old_a=old_a, old_b=old_b)
        if direction == "bwd":
            # producer-fused moments (csrc/prior_hint.h): when the consumer of this message is a separable
            # prior, the kernel that writes the message also leaves sum v_i, and the prior reads it once
            hint = self._moment_hint(i_out)
            if hint is not None:
                self.ctx._call("tramp_prior_hint", hint[0], hint[1], hint[2], hint[3], out_b.shape[1], self.batch,
                               out_b.ptr, out_a.ptr)
        elif factor.n_prev:
            # the same towards a sign / abs likelihood: its input message is the one this channel writes
            hint = self._lik_hint(i_out, out_b.shape[1])
            if hint is not None:
                self.ctx._call("tramp_lik_hint", hint[0], out_b.shape[1], self.batch, out_b.ptr, out_a.ptr, hint[1])
        if factor.n_prev == 0:
            factor.compute_forward_message(a_bwd[i_out], b_bwd[i_out], **kw)
        elif factor.n_next == 0:
            factor.compute_backward_message(self.a_fwd[i_out], self.b_fwd[i_out], **kw)
        elif direction == "fwd":
            factor.compute_forward_message(self.a_fwd[i_in], self.b_fwd[i_in], a_bwd[i_out], b_bwd[i_out], **kw)
        else:
            factor.compute_backward_message(self.a_fwd[i_out], self.b_fwd[i_out],
                                            self.a_bwd[i_in], self.b_bwd[i_in], **kw)
        out_b.version += 1          # consumers (LinearChannel projection cache) see a new message
        if const and not d:
            self._const_done.add((factor, direction, out_b.ptr))

    def _moment_hint(self, i):
        "the prior that consumes the backward message into variable i, if it can use fused moments"
        if i in self._rot:
            return None                 # the stored message lives in a rotated basis
        f = self.model.prev_factor(self.variables[i])
        return f.moment_hint() if f.n_prev == 0 and hasattr(f, "moment_hint") else None

    def _lik_hint(self, i, n):
        "the likelihood that consumes the forward message into variable i, if it can use fused moments"
        if i in self._rot:
            return None
        f = self.model.next_factor(self.variables[i])
        return f.lik_hint(self.batch, n) if f is not None and f.n_next == 0 and hasattr(f, "lik_hint") else None

    def forward_message(self):
        "message_passing.py:249-255"
        if getattr(self, "_adaptive", False):
            return self._adaptive_forward()
        for i, v in enumerate(self.variables):
            self._update(self.model.prev_factor(v), "fwd", i - 1, i)

    def backward_message(self):
        "message_passing.py:257-263"
        if getattr(self, "_adaptive", False):
            return self._adaptive_backward()
        for i in reversed(range(len(self.variables))):
            self._update(self.model.next_factor(self.variables[i]), "bwd", i + 1, i)

    # -- adaptive damping (tramp/algos/message_passing.py:151-185) -------------------------------------------------
    # The reference keeps FOUR messages per SISO variable x between its factors f (before) and g (after):
    #   f -> x and x -> g (direction fwd), g -> x and x -> f (direction bwd).  Without adaptive damping the message a
    # variable passes on is the one it received from the other side, so the chain scheduler stores two; here the
    # pass-through edges are damped on their own (damp_message runs on every edge) and get their own buffers
    # (a_vf / b_vf: x -> g, a_vb / b_vb: x -> f).  Every instance of the ensemble searches its own step beta.
    def _adaptive_allocate(self, fresh):
        ctx, B = self.ctx, self.batch
        if fresh or getattr(self, "a_vf", None) is None:
            self.a_vf = [ctx.empty((B,), np.float64) for _ in self.variables]
            self.a_vb = [ctx.empty((B,), np.float64) for _ in self.variables]
            self.b_vf = [ctx.empty(tuple(b.shape), self.dtype) for b in self.b_fwd]
            self.b_vb = [ctx.empty(tuple(b.shape), self.dtype) for b in self.b_bwd]
            # message_passing.py:211-232: every edge starts from the initializer's value for its direction
            for i in range(len(self.variables)):
                self.a_vf[i].copy_from(self.a_fwd[i])
                self.b_vf[i].copy_from(self.b_fwd[i])
                self.a_vb[i].copy_from(self.a_bwd[i])
                self.b_vb[i].copy_from(self.b_bwd[i])
                self.b_vf[i].version = self.b_vb[i].version = 0
        self._ad_mask = ctx.zeros((4 + B,), np.int32)
        self._ad_out = ctx.empty((B,), np.float64)
        self._ad_new = {}                                    # message length -> (a_new, b_new, a_cand, b_cand)

    def _ad_buffers(self, n):
        if n not in self._ad_new:
            ctx, B = self.ctx, self.batch
            self._ad_new[n] = (ctx.empty((B,), np.float64), ctx.empty((B, n), self.dtype),
                               ctx.empty((B,), np.float64), ctx.empty((B, n), self.dtype))
        return self._ad_new[n]

    def _var_logz(self, a1, b1, a2, b2):
        "Variable.log_partition of two messages (tramp/base.py:146-175), one value per instance"
        self.ctx._call("tramp_log_partition", dtype_code(self.dtype), 9, b1.shape[1], self.batch, 0.0, 0.0, 0.0,
                       a1.ptr, b1.ptr, a2.ptr, b2.ptr, self._ad_out.ptr)
        return self._ad_out.get()

    def _adaptive_edge(self, new_a, new_b, live_a, live_b, objective):
        """compute_adaptive_damping (message_passing.py:151-185) for one edge and every instance: the first
        beta = 1, 1/2, ... 1/512 whose candidate old + beta (new - old) does not lower A_target - A_edge, else the old
        message stays.  ``objective(a, b)`` evaluates that difference with the edge's message replaced."""
        B, n, dt = self.batch, live_b.shape[1], dtype_code(self.dtype)
        if self.n_iter == 0:                                 # :152-153: the first iteration is not damped
            live_a.copy_from(new_a)
            live_b.copy_from(new_b)
            live_b.version = getattr(live_b, "version", 0) + 1
            return
        with np.errstate(all="ignore"):
            A_old = np.atleast_1d(objective(live_a, live_b))
        remaining = np.ones(B, dtype=bool)
        _, _, ca, cb = self._ad_buffers(n)
        for k in range(10):
            ca.copy_from(new_a)
            cb.copy_from(new_b)
            if k:
                # candidate = (1 - beta) old + beta new; instances that already accepted hold their new message in
                # the live buffer - their candidate is meaningless and masked out below
                self.ctx._call("tramp_damp", dt, n, B, 1.0 - 0.5**k, live_a.ptr, live_b.ptr, ca.ptr, cb.ptr)
            with np.errstate(all="ignore"):
                dA = np.atleast_1d(objective(ca, cb)) - A_old
            accept = remaining & (dA >= 0)                   # NaN / inf - inf compare False, as in the reference
            if accept.any():
                mask = np.zeros(4 + B, dtype=np.int32)
                mask[4:] = accept
                self._ad_mask.set(mask)
                self.ctx._call("tramp_ep_restore", dt, n, B, self._ad_mask, cb, live_b, 0)
                self.ctx._call("tramp_ep_restore_a", 1, B, self._ad_mask, ca, live_a, 0)
                remaining &= ~accept
            if not remaining.any():
                break
        live_b.version = getattr(live_b, "version", 0) + 1

    def _factor_logz(self, f, i_var, side, a, b):
        """log-partition of factor f with ONE of its incoming messages replaced by (a, b): side "z" = the message from the
        variable before it (x_{i_var} -> f), side "x" = the message from the variable after it (x_{i_var} -> f)"""
        if f.n_prev == 0:
            return f.compute_log_partition(a, b)
        if f.n_next == 0:
            return f.compute_log_partition(a, b)
        if side == "z":
            return f.compute_log_partition(a, b, self.a_vb[i_var + 1], self.b_vb[i_var + 1])
        return f.compute_log_partition(self.a_vf[i_var - 1], self.b_vf[i_var - 1], a, b)

    def _adaptive_forward(self):
        m = self.model
        for i, v in enumerate(self.variables):
            f, g = m.prev_factor(v), m.next_factor(v)
            na, nb, _, _ = self._ad_buffers(self.b_fwd[i].shape[1])
            # f -> x_i from the messages INTO f: x_{i-1} -> f and x_i -> f
            kw = dict(out_a=na, out_b=nb)
            if f.n_prev == 0:
                f.compute_forward_message(self.a_vb[i], self.b_vb[i], **kw)
            else:
                f.compute_forward_message(self.a_vf[i - 1], self.b_vf[i - 1], self.a_vb[i], self.b_vb[i], **kw)
            # target x_i: A_x(f -> x, g -> x) - A_x(f -> x, x -> f)      (message_passing.py:158-167)
            self._adaptive_edge(na, nb, self.a_fwd[i], self.b_fwd[i],
                                lambda a, b, i=i: self._var_logz(a, b, self.a_bwd[i], self.b_bwd[i])
                                - self._var_logz(a, b, self.a_vb[i], self.b_vb[i]))
            if g is None:
                continue
            # x_i -> g: the variable passes on what it got from f (base.py:183-194); target g, edge pair (x -> g, g -> x)
            self._adaptive_edge(self.a_fwd[i], self.b_fwd[i], self.a_vf[i], self.b_vf[i],
                                lambda a, b, i=i, g=g: np.atleast_1d(self._factor_logz(g, i, "z", a, b))
                                - self._var_logz(a, b, self.a_bwd[i], self.b_bwd[i]))

    def _adaptive_backward(self):
        m = self.model
        for i in reversed(range(len(self.variables))):
            v = self.variables[i]
            f, g = m.prev_factor(v), m.next_factor(v)
            if g is not None:
                na, nb, _, _ = self._ad_buffers(self.b_bwd[i].shape[1])
                kw = dict(out_a=na, out_b=nb)
                if g.n_next == 0:
                    g.compute_backward_message(self.a_vf[i], self.b_vf[i], **kw)
                else:
                    g.compute_backward_message(self.a_vf[i], self.b_vf[i], self.a_vb[i + 1], self.b_vb[i + 1], **kw)
                self._adaptive_edge(na, nb, self.a_bwd[i], self.b_bwd[i],
                                    lambda a, b, i=i: self._var_logz(a, b, self.a_fwd[i], self.b_fwd[i])
                                    - self._var_logz(a, b, self.a_vf[i], self.b_vf[i]))
            # x_i -> f: what came from g; target f, edge pair (x -> f, f -> x)
            self._adaptive_edge(self.a_bwd[i], self.b_bwd[i], self.a_vb[i], self.b_vb[i],
                                lambda a, b, i=i, f=f: np.atleast_1d(self._factor_logz(f, i, "x", a, b))
                                - self._var_logz(a, b, self.a_fwd[i], self.b_fwd[i]))

    def update_variables(self):
        "message_passing.py:265-269 -> Variable.posterior_rv (base.py:157-161), kernel K11"
        dt = dtype_code(self.dtype)
        for i in range(len(self.variables)):
            n = self.r[i].shape[1]
            # reads the previous mean from r, writes the new one to r_prev, then the two swap roles
            self.ctx._call("tramp_variable_posterior", dt, n, self.batch, self.a_fwd[i].ptr,
                           self.b_fwd[i].ptr, self.a_bwd[i].ptr, self.b_bwd[i].ptr, self.r[i].ptr,
                           self.r_prev[i].ptr, self.S.row(3 * i).ptr, self.S.row(3 * i + 1).ptr,
                           self.S.row(3 * i + 2).ptr)
            self.r[i], self.r_prev[i] = self.r_prev[i], self.r[i]

    def _swap_r(self):
        for i in range(len(self.variables)):
            self.r[i], self.r_prev[i] = self.r_prev[i], self.r[i]

    def _monitor_step(self, tol, max_increase, wait_increase, can_stop=True):
        """NaN guard + per-instance stopping rule of the iteration just launched, on the device (monitor.hip).
        can_stop=False (no stopping rule given: tol = -1): only the guard - no instance can ever stop, the
        freeze launches (one per variable) would copy nothing."""
        L = len(self.variables)
        snap = getattr(self, "_snap", None) if self._restore else None
        self.ctx._call("tramp_ep_monitor", L, self._n_edges, self.batch, self.A, self.S, float(tol),
                       float(max_increase), int(wait_increase), self._state, self._v_prev, self._v_frozen,
                       self._a_frozen, snap["A"] if snap else None)
        if not can_stop:
            return
        dt = dtype_code(self.dtype)
        for i in range(L):
            self.ctx._call("tramp_ep_freeze", dt, self.r[i].shape[1], self.batch, self._state, self.r[i],
                           self.r_prev[i], self._frozen_r[i])
        if snap:
            # the MESSAGES an instance that stops now would have left behind: the live ones if it converged, the
            # snapshot taken before this iteration if it gave up (callbacks.py:277-283)
            for live, prev, frozen in self._message_triples():
                self.ctx._call("tramp_ep_freeze", dt, live.shape[1], self.batch, self._state, live, prev, frozen)

    # -- "restoring old message dag" (message_passing.py:187-209, callbacks.py:233-240,277-283)
    _restore = False

    def _message_triples(self):
        s = self._snap
        return ([(b, s["fwd"][i], s["frozen_fwd"][i]) for i, b in enumerate(self.b_fwd)]
                + [(b, s["bwd"][i], s["frozen_bwd"][i]) for i, b in enumerate(self.b_bwd)])

    # The chain scheduler does not copy: it alternates two sets of message buffers (b vectors and the precision table),
    # so the set written in the previous iteration is the snapshot (`_snap["fwd"], ["bwd"], ["A"]` ARE that set).
    # tools/bench_restore_overhead.py: the four copies cost +16 % per iteration at 10^6 x 64.  The tree scheduler
    # (tree.py) alternates its buffer sets the same way; a scheduler that cannot sets _pingpong = False and gets the
    # copying branch of _take_snapshot.
    _pingpong = True

    def _pp_active(self):
        return bool(self._restore and self._pingpong and getattr(self, "_snap", None) is not None)

    def _allocate_snapshot(self):
        ctx, L = self.ctx, len(self.variables)

        def mk(bufs):
            out = [ctx.zeros(tuple(b.shape), self.dtype) for b in bufs]
            for b in out:
                b.version = 0
            return out
        A2 = ctx.zeros(tuple(self.A.shape), np.float64)
        self._snap = dict(fwd=mk(self.b_fwd), bwd=mk(self.b_bwd), frozen_fwd=mk(self.b_fwd), frozen_bwd=mk(self.b_bwd), A=A2,
                          a_fwd=[A2.row(2 * i) for i in range(L)], a_bwd=[A2.row(2 * i + 1) for i in range(L)])

    def _sync_snapshot(self):
        "both sets hold the current messages (start of a run: the first forward sweep reads the previous set's backward ones)"
        s = self._snap
        for live, prev, _ in self._message_triples():
            prev.copy_from(live)
            prev.version = getattr(prev, "version", 0) + 1
        s["A"].copy_from(self.A)

    def _swap_sets(self):
        s = self._snap
        self.b_fwd, s["fwd"] = s["fwd"], self.b_fwd
        self.b_bwd, s["bwd"] = s["bwd"], self.b_bwd
        self.A, s["A"] = s["A"], self.A
        self.a_fwd, s["a_fwd"] = s["a_fwd"], self.a_fwd
        self.a_bwd, s["a_bwd"] = s["a_bwd"], self.a_bwd

    def _take_snapshot(self):
        """the message dag as it stands before an iteration (the reference deep-copies it every iteration): here the set
        of buffers just written becomes the snapshot and the other set receives this iteration's messages"""
        if self._pingpong:
            self._swap_sets()
            return
        s = self._snap
        for live, prev, _ in self._message_triples():
            prev.copy_from(live)
        s["A"].copy_from(self.A)

    def _restore_messages(self, everybody=False):
        """every stopped instance gets the messages of ITS stopping iteration back (everybody=True: all instances
        go back to the snapshot - a NaN was found); consumers' cached projections of the old buffers are void"""
        dt = dtype_code(self.dtype)
        s = self._snap
        for live, prev, frozen in self._message_triples():
            self.ctx._call("tramp_ep_restore", dt, live.shape[1], self.batch, self._state,
                           prev if everybody else frozen, live, int(everybody))
            live.version += 1
        self.ctx._call("tramp_ep_restore_a", self._n_edges, self.batch, self._state,
                       s["A"] if everybody else self._a_frozen, self.A, int(everybody))
        self.ctx._call("tramp_prior_hint", 0, 0.0, 0.0, 0.0, 0, 0, None, None)
        self._const_done = set()

    def _poll_monitor(self):
        """Read the state words; raises on a NaN message (message_passing.py:187-209), otherwise updates
        status / n_iter_instance / frozen scalars and returns the number of instances still running."""
        st = self._state.get()
        if st[2]:
            e, b = divmod(int(st[2]) - 1, self.batch)
            if self._restore:
                logger.warning("restoring old message dag")
                self._restore_messages(everybody=True)
            raise ValueError(self._edge_name(e) + f" (instance {b}): a is nan")
        B = self.batch
        status, n_it = st[4:4 + B], st[4 + B:4 + 2 * B]
        newly = (status != 0) & (self.status == 0)
        if newly.any():
            self._frozen_v[:, newly] = self._v_frozen.get()[:, newly]
            self._frozen_A[:, newly] = self._a_frozen.get()[:, newly]
            self.status[newly] = status[newly]
            self.n_iter_instance[newly] = getattr(self, "_n0", 0) + n_it[newly]
        return int(st[1])

    def _edge_name(self, e):
        v = self.variables[e // 2]
        return f"{'fwd' if e % 2 == 0 else 'bwd'} message into {v.id}"

    def _read_scalars(self):
        self._S_prev_host = getattr(self, "S_host", None)
        self.A_host = self.A.get()
        self.S_host = self.S.get()

    def check_message(self):
        "message_passing.py:187-209 (a non-finite mean poisons the reduced variance, so a suffices)"
        bad = np.isnan(self.A_host)
        if bad.any():
            e, b = np.argwhere(bad)[0]
            raise ValueError(self._edge_name(int(e)) + f" (instance {b}): a is nan")
        if (self.A_host < 0).any():
            logger.warning("negative a in messages")

    # -- results
    def relative_change(self, ids="all"):
        rows = [i for i, v in enumerate(self.variables) if ids == "all" or v.id in ids]
        sd = np.stack([self.S_host[3 * i + 1] for i in rows])
        sn = np.stack([self.S_host[3 * i + 2] for i in rows])
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.sqrt(sd) / np.sqrt(sn)

    def mark_converged(self, mask):
        """Host-side twin of the device monitor, for callbacks that evaluate their own rule on the host
        (e.g. EarlyStoppingEP on a subset of the variables): record the instances that meet the rule for
        the first time and freeze their result.  The reference would have stopped each instance's own
        run here; the ensemble keeps iterating until all are done (instances never interact)."""
        newly = mask & (self.status == 0)
        if not newly.any():
            return
        self.n_iter_instance[newly] = self.n_iter
        self.status[newly] = 1
        self._frozen_A[:, newly] = self.A_host[:, newly]
        for b in np.flatnonzero(newly):
            for i in range(len(self.variables)):
                self._frozen_r[i].row(int(b)).copy_from(self.r[i].row(int(b)))
                self._frozen_v[i, b] = self.S_host[3 * i, b]

    def mark_diverged(self, mask):
        """An instance whose relative change exceeds max_increase gives up: it reports the iterate BEFORE the
        increase (the reference restores its previous message dag, callbacks.py:277-283).  The previous
        means are at hand where the variable-posterior kernel ping-pongs (chain scheduler); the previous
        variances when the scalar table was read every iteration."""
        newly = mask & (self.status == 0)
        if not newly.any():
            return
        self.n_iter_instance[newly] = self.n_iter
        self.status[newly] = 2
        self._frozen_A[:, newly] = self.A_host[:, newly]
        prev_r = getattr(self, "r_prev", None) if self._device_monitor else None
        S_prev = self._S_prev_host if getattr(self, "_S_prev_host", None) is not None else self.S_host
        for b in np.flatnonzero(newly):
            for i in range(len(self.variables)):
                src = prev_r[i] if prev_r is not None else self.r[i]
                self._frozen_r[i].row(int(b)).copy_from(src.row(int(b)))
                self._frozen_v[i, b] = S_prev[3 * i, b]

    def get_variables_data(self, ids="all", want_r=True, frozen=None):
        """{id: dict(r=..., v=...)}.  An instance that has stopped (converged, or given up after an
        increase: ``self.status``) reports the estimate of ITS stopping iteration - for a diverged one
        the iterate before the increase, as the reference restores its previous message dag
        (callbacks.py:277-283) - while the ensemble keeps iterating for the others.  ``frozen=False``
        returns the current iterate of every instance."""
        out = {}
        done = self.status != 0
        use_frozen = (frozen is None or frozen) and done.any()
        for i, v in enumerate(self.variables):
            if ids == "all" or v.id in ids:
                vv = self.S_host[3 * i].copy()
                if use_frozen:
                    vv[done] = self._frozen_v[i, done]
                d = dict(v=float(vv[0]) if self.batch == 1 else vv)
                if want_r:
                    unrot = (lambda a: self._rotate(i, a, back=True)) if i in self._rot else (lambda a: a)
                    if use_frozen and done.all():
                        r = unrot(self._frozen_r[i]).get()
                    else:
                        r = unrot(self.r[i]).get()
                        if use_frozen:
                            r[done] = unrot(self._frozen_r[i]).get()[done]
                    # in the variable's own shape, as the reference hands estimates out ((2, n) planar complex, images, (d,) + shape
                    # gradient fields); an ensemble keeps its leading instance axis
                    shape = tuple(self.shapes[v.id])
                    d["r"] = r[0].reshape(shape) if self.batch == 1 else r.reshape((self.batch,) + shape)
                out[v.id] = d
        return out

    def get_variable_data(self, id):
        return self.get_variables_data([id])[id]

    def get_edges_data(self):
        "a of every factor->variable edge, as {(variable id, direction): array[batch]}"
        out = {}
        A = self.A_host.copy()
        done = self.status != 0
        A[:, done] = self._frozen_A[:, done]                  # a stopped instance: the values of its last iteration
        for i, v in enumerate(self.variables):
            out[(v.id, "fwd")] = A[2 * i]
            out[(v.id, "bwd")] = A[2 * i + 1]
        return out

    def get_message(self, id, direction):
        i = self.variable_index(id)
        a = (self.a_fwd if direction == "fwd" else self.a_bwd)[i].get()
        b = (self.b_fwd if direction == "fwd" else self.b_bwd)[i]
        b = (self._rotate(i, b, back=True) if i in self._rot else b).get()
        return (float(a[0]), b[0]) if self.batch == 1 else (a, b)

    def variable_index(self, id):
        return [v.id for v in self.variables].index(id)

    def mse(self, id, x_true, frozen=None):
        """mean squared error of the posterior mean against ``x_true``: [n] (one signal shared by the ensemble)
        or [batch, n].  As in get_variables_data, an instance that has stopped reports the estimate of ITS
        stopping iteration (for one that gave up: the iterate before the increase); ``frozen=False`` scores the
        current iterate of every instance."""
        i = self.variable_index(id)
        done = self.status != 0
        use_frozen = (frozen is None or frozen) and done.any()
        src = self.r[i]
        if use_frozen:
            if done.all():
                src = self._frozen_r[i]
            else:
                src = self.ctx.empty(tuple(self.r[i].shape), self.dtype)
                src.copy_from(self.r[i])
                for b in np.flatnonzero(done):
                    src.row(int(b)).copy_from(self._frozen_r[i].row(int(b)))
        r = self._rotate(i, src, back=True) if i in self._rot else src
        n = r.shape[1]
        x = np.asarray(x_true, dtype=self.dtype)
        if x.size == n:
            x = np.broadcast_to(x.reshape(1, n), (self.batch, n))
        elif x.size == self.batch * n and x.shape[0] == self.batch:
            x = x.reshape(self.batch, n)
        else:
            raise ValueError(f"x_true for {id!r} must have shape ({n},) or ({self.batch}, {n}), got {x.shape}")
        xt = self.ctx.array(np.ascontiguousarray(x), self.dtype)
        out = self.ctx.empty((self.batch,), np.float64)
        self.ctx._call("tramp_sq_dist", dtype_code(self.dtype), n, self.batch, r.ptr, xt.ptr, out.ptr)
        m = out.get() / n
        return float(m[0]) if self.batch == 1 else m

    def log_evidence(self):
        """EP log-evidence A_model (tramp/algos/expectation_propagation.py:21-27,
        tramp/algos/message_passing.py:306-328): node log-partitions minus the fwd-edge ones.  On a
        chain of SISO variables both edges of a variable carry the variable's own (a, b) pair, so
        this is sum_factors A_f - sum_variables A_x.  One value per instance."""
        active = getattr(self, "_basis_depth", 0) > 0
        if active:
            self._apply_basis(False)                            # the formula below runs in the original basis
        saved = (self.b_fwd, self.b_bwd)
        try:
            self.b_fwd = [self._rotate(i, b, back=True) if i in self._rot else b for i, b in enumerate(self.b_fwd)]
            self.b_bwd = [self._rotate(i, b, back=True) if i in self._rot else b for i, b in enumerate(self.b_bwd)]
            return self._log_evidence()
        finally:
            self.b_fwd, self.b_bwd = saved
            if active:
                self._apply_basis(True)

    def _log_evidence(self):
        L = len(self.variables)
        A = np.zeros(self.batch)
        for i, v in enumerate(self.variables):
            f = self.model.prev_factor(v)
            if f.n_prev == 0:
                A = A + np.atleast_1d(f.compute_log_partition(self.a_bwd[i], self.b_bwd[i]))
            else:
                A = A + np.atleast_1d(f.compute_log_partition(self.a_fwd[i - 1], self.b_fwd[i - 1],
                                                              self.a_bwd[i], self.b_bwd[i]))
        last = self.model.next_factor(self.variables[-1])
        if last is not None:
            A = A + np.atleast_1d(last.compute_log_partition(self.a_fwd[L - 1], self.b_fwd[L - 1]))
        out = self.ctx.empty((self.batch,), np.float64)
        for i in range(L):
            n = self.b_fwd[i].shape[1]
            self.ctx._call("tramp_log_partition", dtype_code(self.dtype), 9, n, self.batch, 0.0, 0.0, 0.0,
                           self.a_fwd[i].ptr, self.b_fwd[i].ptr, self.a_bwd[i].ptr, self.b_bwd[i].ptr, out.ptr)
            A = A - out.get()
        return float(A[0]) if self.batch == 1 else A

    def surprisal(self):
        return -self.log_evidence()                        # expectation_propagation.py:29-32

    # -- main loop
    _device_monitor = True       # chain scheduler: stopping rule and NaN guard run on the device (monitor.hip)

    def _one_iteration(self, mon):
        if self._restore:
            self._take_snapshot()
        self.forward_message()
        self.backward_message()
        self.update_variables()
        if mon is not None:
            self._monitor_step(*mon)

    def _capture_iterations(self, mon, count):
        """Record `count` steady-state iterations (every launch of the three sweeps and of the monitor) into
        one hipGraph.  Two when the monitor is on: the variable-posterior kernel ping-pongs between r and
        r_prev, so the launch sequence repeats with period two."""
        import ctypes as C
        self.ctx._call("tramp_graph_begin")
        try:
            for _ in range(count):
                self._one_iteration(mon)
        finally:
            g = C.c_void_p()
            self.ctx._call("tramp_graph_end", C.byref(g))
        return g

    @staticmethod
    def _split_callback(callback):
        """(stopping rule the device monitor can run, callbacks that stay on the host).  The monitor takes an
        EarlyStoppingEP over all variables (also inside a JoinCallback); anything else - subclasses,
        wrappers, a rule on a subset of the variables - keeps its host-side evaluation."""
        members = list(callback.callbacks) if type(callback) is JoinCallback else [callback]
        stop = [cb for cb in members if type(cb) is EarlyStoppingEP and cb.ids == "all"]
        if len(stop) != 1:
            return None, members
        return stop[0], [cb for cb in members if cb is not stop[0] and type(cb) is not PassCallback]

    def iterate(self, max_iter=200, callback=None, initializer=None, damping=None, warm_start=False,
                check_every=1, graph=False, restore_messages=None, update_dA=False):
        """message_passing.py:330-357.  The NaN guard and the per-instance stopping rule run on the device
        after every iteration (csrc/monitor.hip); ``check_every`` = k makes the host look at their state
        words - and run the remaining host callbacks - only every k-th iteration, the sweeps themselves
        never wait on the host.  An instance that stops keeps the estimate of its own stopping iteration
        (``status``, ``n_iter_instance``, get_variables_data) while the others go on; k > 1 only delays the
        moment the loop notices that all are done.  ``graph=True`` captures the steady-state launch sequence
        in a hipGraph and replays it (launch-bound small problems); results are identical to the eager loop.
        ``restore_messages`` (default: on whenever the device runs a stopping rule; it keeps a second set of message
        buffers and a frozen set per edge - three times the message memory, about 3 GB per variable at 10^6 x 64 f64 -
        and one freeze kernel per message and iteration; pass False where neither fits): the reference's "restoring
        old message dag" per instance - the messages are snapshot before every iteration (two alternating sets of buffers, no copies),
        an instance that gives up after an increase gets the previous iteration's messages back, a converged one
        keeps those of its stopping iteration, and a NaN puts every instance back one iteration before ValueError
        is raised (message_passing.py:187-209, callbacks.py:233-240,277-283); a warm start then continues each
        instance from where its own run would have stopped."""
        if update_dA:
            raise NotImplementedError("update_dA (the per-message free-energy increments of message_passing.py:245-246) is not tracked")
        initializer = initializer or ConstantInit(a=0, b=0)
        callback = callback or self.default_stopping
        adaptive = isinstance(damping, str) and damping == "adaptive"
        if isinstance(damping, (int, np.integer, np.floating)) and not isinstance(damping, bool):
            damping = float(damping)                           # the reference's examples pass damping=0
        if damping is not None and damping is not False and not isinstance(damping, float) and not adaptive:
            raise ValueError("damping must be None, a float (constant damping) or 'adaptive'")
        self.damping = 0.0 if adaptive else (float(damping) if damping else 0.0)
        check_every = max(int(check_every), 1)
        if adaptive:
            # message_passing.py:151-185: a step-halving search on the node objectives, per edge - host-orchestrated here
            # (every candidate costs log-partition evaluations whose per-instance values come back to the host)
            if graph:
                raise ValueError("adaptive damping cannot be captured in a hipGraph")
            if not getattr(self.model, "is_chain", True):
                raise NotImplementedError("adaptive damping is implemented for chain models (SISO variables)")
            if self._rot and (warm_start or self._allocated):
                if warm_start:
                    raise ValueError("adaptive damping cannot warm-start a run that carried variables in a rotated basis")
            if self._rot:
                self._rot_saved = self._rot                  # a later non-adaptive run gets its rotated bases back
                self._rot = {}                               # all messages in the original basis
                self._allocated = False
        elif getattr(self, "_rot_saved", None) and not warm_start:
            # back from an adaptive run: the rotated-basis fast path again, and no stale variable -> factor buffers
            self._rot, self._rot_saved = self._rot_saved, None
            self._allocated = False
            for nm in ("a_vf", "b_vf", "a_vb", "b_vb"):
                if hasattr(self, nm):
                    delattr(self, nm)
        self._adaptive = adaptive
        if warm_start:
            if not self._allocated:
                raise ValueError("message dag was never initialized")
        else:
            self.init_message_dag(initializer)
            self.n_iter = 0
        if adaptive:
            # a warm start after a NON-adaptive run: the variable -> factor buffers are from before it, rebuild them
            self._adaptive_allocate(fresh=(not warm_start) or not getattr(self, "_prev_adaptive", False))
        self._prev_adaptive = adaptive
        if not adaptive:
            self._auto_attach_digits()
        if self._device_monitor and not adaptive:
            self._reset_monitor()            # a warm start begins a new run of the stopping rule (i = 0 again)
            stopper, host_cbs = self._split_callback(callback)
            mon = ((stopper.tol, stopper.max_increase, stopper.wait_increase, True) if stopper is not None
                   else (-1.0, np.inf, 0, False))
        else:
            stopper, host_cbs, mon = None, [callback], None
            self.status[:] = 0
            self.n_iter_instance[:] = 0
        self._restore = bool(self._device_monitor and not adaptive and
                             (restore_messages if restore_messages is not None else stopper is not None))
        if self._restore and getattr(self, "_snap", None) is None:
            self._allocate_snapshot()
        n0 = self._n0 = self.n_iter
        period = 2 if mon is not None else 1           # r / r_prev ping-pong: the launch sequence has period two
        graphs = {}
        basis = self._Basis(self)
        basis.__enter__()
        if self._pp_active():
            self._sync_snapshot()
        try:
            i = 0
            while i < max_iter:
                par = i % period
                if graph and i >= 2:
                    # iterations 0 and 1 ran eagerly: buffers exist, caches are in steady state
                    if par not in graphs:
                        graphs[par] = self._capture_iterations(mon, 1)     # (the Python-side swaps happen here)
                        self.ctx._call("tramp_graph_launch", graphs[par])
                    else:
                        if self._pp_active():
                            self._swap_sets()            # what _take_snapshot did when this iteration was captured
                        self.ctx._call("tramp_graph_launch", graphs[par])
                        if mon is not None:
                            self._swap_r()
                else:
                    self._one_iteration(mon)
                step = 1
                first, i = i, i + step
                self.n_iter += step
                # poll when a multiple of check_every was crossed, and at the end
                if i // check_every == first // check_every and i < max_iter:
                    continue
                self._read_scalars()
                running = None
                if mon is not None:
                    running = self._poll_monitor()
                else:
                    self.check_message()
                stop = False
                for cb in host_cbs:
                    stop = bool(cb(self, i - 1, max_iter)) or stop
                if stopper is not None and running == 0:
                    logger.info("early stopping: every instance has met the rule (tol=%.2e) or given up", stopper.tol)
                    stop = True
                if stop:
                    logger.info("terminated after n_iter=%d iterations", self.n_iter)
                    break
        finally:
            basis.__exit__(None, None, None)
            for g in graphs.values():
                self.ctx._call("tramp_graph_destroy", g)
        if self._restore and stopper is not None and self.status.any():
            self._restore_messages()
        if stopper is not None and self.status.all():
            self.n_iter = int(self.n_iter_instance.max())          # the iteration the last instance stopped at
        self.n_iter_instance[self.status == 0] = self.n_iter       # still running: cut short by max_iter
        return self


# ------------------------------------------------------------------ SE
class StateEvolution:
    """Scalar state evolution on the same chain (tramp/algos/state_evolution.py:5-27).

    Messages are host scalars ``a``; the factor errors are closed forms or the
    device quadratures K13/K14 (factor.compute_*_state_evolution)."""

    def __new__(cls, model, *args, **kwargs):
        if cls is StateEvolution and not getattr(model, "is_chain", True):
            from .tree import TreeStateEvolution
            return TreeStateEvolution(model)
        return super().__new__(cls)

    def __init__(self, model):
        self.model = model
        model.init_second_moments()
        self.variables = model.variables
        self.taus = model.get_second_moments()
        self.n_iter = 0
        self.default_stopping = EarlyStopping()

    def iterate(self, max_iter=200, callback=None, initializer=None, damping=None, warm_start=False, update_dA=False):
        if update_dA:
            raise NotImplementedError("update_dA is not tracked")
        initializer = initializer or ConstantInit(a=0, b=0)
        callback = callback or self.default_stopping
        d = float(damping) if damping else 0.0
        L = len(self.variables)
        if not warm_start:
            self.a_fwd = [initializer.init_a(None, v.id, "fwd") for v in self.variables]
            self.a_bwd = [initializer.init_a(None, v.id, "bwd") for v in self.variables]
            self.n_iter = 0
        m = self.model
        for it in range(max_iter):
            for i, v in enumerate(self.variables):
                f = m.prev_factor(v)
                if f.n_prev == 0:
                    new = f.compute_forward_state_evolution(self.a_bwd[i])
                else:
                    tau = self.taus[self.variables[i - 1].id]
                    new = f.compute_forward_state_evolution(self.a_fwd[i - 1], self.a_bwd[i], tau)
                self._check(new, f"{f.id}->{v.id}")
                self.a_fwd[i] = d * self.a_fwd[i] + (1 - d) * new if d else new
            for i in reversed(range(L)):
                v = self.variables[i]
                f = m.next_factor(v)
                if f is None:
                    continue
                tau = self.taus[v.id]
                if f.n_next == 0:
                    new = f.compute_backward_state_evolution(self.a_fwd[i], tau)
                else:
                    new = f.compute_backward_state_evolution(self.a_fwd[i], self.a_bwd[i + 1], tau)
                self._check(new, f"{f.id}->{v.id}")
                self.a_bwd[i] = d * self.a_bwd[i] + (1 - d) * new if d else new
            self.v = [1. / (self.a_fwd[i] + self.a_bwd[i]) for i in range(L)]     # base.py:167-170
            self.n_iter += 1
            if callback(self, it, max_iter):
                break
        return self

    @staticmethod
    def _check(a, where):
        if np.any(np.isnan(a)):
            raise ValueError(f"{where} a is nan")

    def get_variables_data(self, ids="all", want_r=False):
        return {v.id: dict(v=self.v[i]) for i, v in enumerate(self.variables)
                if ids == "all" or v.id in ids}

    def get_variable_data(self, id):
        return self.get_variables_data([id])[id]

    def get_edges_data(self):
        out = {}
        for i, v in enumerate(self.variables):
            out[(v.id, "fwd")] = self.a_fwd[i]
            out[(v.id, "bwd")] = self.a_bwd[i]
        return out


__all__ = ["ExpectationPropagation", "StateEvolution", "EarlyStoppingEP", "EarlyStopping", "ConstantInit",
           "CustomInit", "NoisyInit", "JoinCallback", "PassCallback", "TrackEvolution", "TrackErrors", "Factor", "Variable",
           "_lib"]
