"""Channels (1-in / 1-out factors): host mirror of tramp.channels for the hot path.

tramp/channels/base_channel.py, channels/linear/linear_channel.py,
channels/linear/analytical_linear_channel.py,
channels/activation/piecewise_linear_channel.py, channels/noise/gaussian_channel.py.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .base import Factor, VecArg, ScalArg, out_vec, ret_vec, ret_scal
from .device import DeviceArray, default_context
from .ensembles import MarchenkoPasturEnsemble


class Channel(Factor):
    "tramp/channels/base_channel.py:5-97"
    n_next = 1
    n_prev = 1

    def compute_forward_state_evolution(self, az, ax, tau_z):
        return self.compute_a_new(self.compute_forward_error(az, ax, tau_z), ax)      # :19-22

    def compute_backward_state_evolution(self, az, ax, tau_z):
        return self.compute_a_new(self.compute_backward_error(az, ax, tau_z), az)     # :24-27

    def compute_forward_overlap(self, az, ax, tau_z):
        return self.second_moment(tau_z) - self.compute_forward_error(az, ax, tau_z)

    def compute_backward_overlap(self, az, ax, tau_z):
        return tau_z - self.compute_backward_error(az, ax, tau_z)


class GaussianChannel(Channel):
    """Additive Gaussian noise channel (tramp/channels/noise/gaussian_channel.py).

    Only the closed-form scalar gains are needed on the hot path: once its output is
    observed the model builder turns it into a GaussianLikelihood
    (tramp/models/dag_algebra.py:22-23).
    """

    def __init__(self, var=1, dtype=np.float64, ctx=None):
        self.var = var
        self.repr_init()
        self._init_device(dtype, ctx)
        self.sigma = np.sqrt(var)
        self.a = 1 / var

    def sample(self, Z):
        return Z + self.sigma * np.random.standard_normal(Z.shape)

    def second_moment(self, tau_z):
        return tau_z + self.var

    def _message(self, a_in, b_in, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None):
        "as a hidden factor (gaussian_channel.py:23-33): the message shrinks by k = a / (a + a_in)"
        ctx = self.ctx
        b = VecArg(ctx, b_in, self.dtype)
        a = ScalArg(ctx, a_in, b.batch)
        a_new = out_a if out_a is not None else ctx.empty((b.batch,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, b, self.dtype)
        ctx._call("tramp_gauss_channel", self._dt, b.n, b.batch, float(self.var), a.dev.ptr, b.dev.ptr,
                  float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr)
        return ret_scal(a_new, a, b), ret_vec(b_new, b)

    def compute_forward_message(self, az, bz, ax, bx, **kw):
        return self._message(az, bz, **kw)

    def compute_backward_message(self, az, bz, ax, bx, **kw):
        return self._message(ax, bx, **kw)

    def compute_forward_state_evolution(self, az, ax, tau_z):
        return self.a / (self.a + az) * az                     # gaussian_channel.py:35-38

    def compute_backward_state_evolution(self, az, ax, tau_z):
        return self.a / (self.a + ax) * ax                     # :40-43


class BiasChannel(Channel):
    """x = z + bias (tramp/channels/linear/bias_channel.py:5-33): the precision passes through, the mean
    message is shifted by a * bias.  One bias vector shared by the ensemble."""

    def __init__(self, bias, dtype=np.float64, ctx=None):
        self.repr_init()
        self._init_device(dtype, ctx)
        self.bias = np.asarray(bias, dtype=np.float64)
        self._bias_dev = None

    def sample(self, Z):
        return Z + self.bias

    def second_moment(self, tau_z):
        return tau_z + (self.bias**2).mean()                   # :16-18

    def compute_forward_state_evolution(self, az, ax, tau_z):
        return az                                              # :31-33

    def compute_backward_state_evolution(self, az, ax, tau_z):
        return ax                                              # :35-37

    def _message(self, direction, a_in, b_in, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None):
        ctx = self.ctx
        if self._bias_dev is None:
            self._bias_dev = ctx.array(self.bias.reshape(-1).astype(self.dtype), self.dtype)
        b = VecArg(ctx, b_in, self.dtype)
        if b.n != self._bias_dev.shape[0]:
            raise ValueError(f"expected messages of {self._bias_dev.shape[0]} components, got {b.n}")
        a = ScalArg(ctx, a_in, b.batch)
        a_new = out_a if out_a is not None else ctx.empty((b.batch,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, b, self.dtype)
        ctx._call("tramp_bias_channel", self._dt, direction, b.n, b.batch, self._bias_dev.ptr, a.dev.ptr, b.dev.ptr,
                  float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr)
        return ret_scal(a_new, a, b), ret_vec(b_new, b)

    def compute_forward_message(self, az, bz, ax, bx, **kw):
        return self._message(_lib.FWD, az, bz, **kw)            # :21-24

    def compute_backward_message(self, az, bz, ax, bx, **kw):
        return self._message(_lib.BWD, ax, bx, **kw)            # :26-29


class SumChannel(Channel):
    """x = z_1 + ... + z_K (tramp/channels/linear/sum_channel.py:5-56): the K-input factor of the committee
    machine (tramp/models/committee_model.py:35).  ``az``, ``bz`` are lists of K messages."""

    n_next = 1

    def __init__(self, n_prev, dtype=np.float64, ctx=None):
        self.n_prev = n_prev
        self.repr_init()
        self._init_device(dtype, ctx)

    def sample(self, *Zs):
        if len(Zs) != self.n_prev:
            raise ValueError(f"expect {self.n_prev} arrays")
        return sum(Zs)

    def second_moment(self, *tau_zs):
        if len(tau_zs) != self.n_prev:
            raise ValueError(f"expect {self.n_prev} arrays")
        return sum(tau_zs)

    def compute_forward_state_evolution(self, az, ax, tau_z):
        return 1 / sum(1 / a for a in az)                                    # :45-49

    def compute_backward_state_evolution(self, az, ax, tau_z):
        v_bar = sum(1 / a for a in az)                                       # :51-56
        return [1 / (1 / ax + v_bar - 1 / a) for a in az]

    def _message(self, target, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None):
        ctx = self.ctx
        if len(az) != self.n_prev or len(bz) != self.n_prev:
            raise ValueError(f"expect {self.n_prev} messages")
        zs = [VecArg(ctx, b, self.dtype) for b in bz]
        B, n = zs[0].batch, zs[0].n
        if any((z.batch, z.n) != (B, n) for z in zs):
            raise ValueError("the summed messages must have the same shape")
        azs = [ScalArg(ctx, a, B) for a in az]
        ap = (C.c_void_p * self.n_prev)(*[a.dev.ptr for a in azs])
        bp = (C.c_void_p * self.n_prev)(*[z.dev.ptr for z in zs])
        x = VecArg(ctx, bx, self.dtype) if target >= 0 else None
        a_x = ScalArg(ctx, ax, B) if target >= 0 else None
        a_new = out_a if out_a is not None else ctx.empty((B,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, zs[0], self.dtype)
        ctx._call("tramp_sum_channel", self._dt, n, B, self.n_prev, ap, bp,
                  a_x.dev.ptr if a_x is not None else None, x.dev.ptr if x is not None else None, target,
                  float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr)
        return ret_scal(a_new, azs[0], zs[0]), ret_vec(b_new, zs[0])

    def compute_forward_message(self, az, bz, ax, bx, **kw):
        return self._message(-1, az, bz, ax, bx, **kw)                        # :26-32

    def compute_backward_message(self, az, bz, ax, bx, target=None, **kw):
        "message to z_target; without ``target`` the list of all K messages as the reference returns (:34-43)"
        if target is not None:
            return self._message(target, az, bz, ax, bx, **kw)
        outs = [self._message(k, az, bz, ax, bx) for k in range(self.n_prev)]
        return [o[0] for o in outs], [o[1] for o in outs]


class DuplicateChannel(Channel):
    """x_1 = ... = x_K = z (tramp/channels/shape/duplicate_channel.py:4-51): one input, K outputs.  ``ax``, ``bx`` are lists
    of K messages; the forward message is the list of K messages, or the one to ``x_target``."""

    n_prev = 1

    def __init__(self, n_next, dtype=np.float64, ctx=None):
        self.n_next = n_next
        self.repr_init()
        self._init_device(dtype, ctx)

    def sample(self, Z):
        return (Z,) * self.n_next

    def second_moment(self, tau_z):
        return (tau_z,) * self.n_next

    def compute_backward_error(self, az, ax, tau_z):
        return 1.0 / (az + sum(ax))                                           # :39-42

    def compute_forward_error(self, az, ax, tau_z):
        return [self.compute_backward_error(az, ax, tau_z)] * self.n_next     # :34-37

    def compute_backward_state_evolution(self, az, ax, tau_z):
        v = self.compute_backward_error(az, ax, tau_z)                        # base_channel.py:107-110
        return np.clip(1.0 / v - az, 1e-11, 1e11)

    def compute_forward_state_evolution(self, az, ax, tau_z):
        v = self.compute_backward_error(az, ax, tau_z)
        return [np.clip(1.0 / v - a, 1e-11, 1e11) for a in ax]

    def _message(self, target, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None):
        ctx = self.ctx
        if len(ax) != self.n_next or len(bx) != self.n_next:
            raise ValueError(f"expect {self.n_next} messages")
        z = VecArg(ctx, bz, self.dtype)
        xs = [VecArg(ctx, b, self.dtype) for b in bx]
        B, n = z.batch, z.n
        if any((x.batch, x.n) != (B, n) for x in xs):
            raise ValueError("the duplicated messages must have the shape of the input")
        a_z = ScalArg(ctx, az, B)
        axs = [ScalArg(ctx, a, B) for a in ax]
        ap = (C.c_void_p * self.n_next)(*[a.dev.ptr for a in axs])
        bp = (C.c_void_p * self.n_next)(*[x.dev.ptr for x in xs])
        a_new = out_a if out_a is not None else ctx.empty((B,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, z, self.dtype)
        ctx._call("tramp_duplicate_channel", self._dt, n, B, self.n_next, a_z.dev.ptr, z.dev.ptr, ap, bp, target,
                  float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr)
        return ret_scal(a_new, a_z, z), ret_vec(b_new, z)

    def compute_backward_message(self, az, bz, ax, bx, **kw):
        return self._message(-1, az, bz, ax, bx, **kw)                        # base_channel.py:102-105

    def compute_forward_message(self, az, bz, ax, bx, target=None, **kw):
        "message to x_target; without ``target`` the list of all K messages as the reference returns (base.py:329-346)"
        if target is not None:
            return self._message(target, az, bz, ax, bx, **kw)
        outs = [self._message(k, az, bz, ax, bx) for k in range(self.n_next)]
        return [o[0] for o in outs], [o[1] for o in outs]

    def compute_backward_posterior(self, az, bz, ax, bx):
        "r = b / a, v = 1 / a with a = az + sum ax, b = bz + sum bx (:24-31) - through the message: r = (b_new + bz) / (az + a_new)"
        a_new, b_new = self.compute_backward_message(az, bz, ax, bx)
        a = np.asarray(az, dtype=float) + np.asarray(a_new, dtype=float)
        r = (np.asarray(b_new, dtype=float) + np.asarray(bz, dtype=float)) / (a[..., None] if np.ndim(a) else a)
        return r, 1.0 / a

    def compute_forward_posterior(self, az, bz, ax, bx):
        rz, vz = self.compute_backward_posterior(az, bz, ax, bx)              # :18-22
        return [rz] * self.n_next, [vz] * self.n_next


# ------------------------------------------------------------------ componentwise
class PiecewiseLinearChannel(Channel):
    """x = f(z), f piecewise linear with two half-line regions (kernel K7).

    tramp/channels/activation/piecewise_linear_channel.py:7-69; region tables of
    SgnChannel :80-84, AbsChannel :87-91, ReluChannel :102-106.
    """
    _kind = None
    _f = None

    def __init__(self, dtype=np.float64, ctx=None):
        self.name = {_lib.SGN: "sgn", _lib.ABS: "abs", _lib.RELU: "relu"}[self._kind]
        self.repr_init()
        self._init_device(dtype, ctx)

    def sample(self, Z):
        return type(self)._f(Z)

    def second_moment(self, tau_z):
        "piecewise_linear_channel.py:21-25 with the half-line moments of N(0, tau_z)"
        # both regions have probability 1/2; |z| has mean sqrt(2 tau/pi), E z^2 = tau
        if self._kind == _lib.SGN:
            return 1.0
        if self._kind == _lib.ABS:
            return tau_z
        return 0.5 * tau_z

    def _run(self, direction, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
             old_b=None, want_posterior=False, want_mask=False):
        ctx = self.ctx
        z = VecArg(ctx, bz, self.dtype)
        x = VecArg(ctx, bx, self.dtype)
        if (z.batch, z.n) != (x.batch, x.n):
            raise ValueError(f"bz {z.dev.shape} and bx {x.dev.shape} must have the same shape")
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        a_new = out_a if out_a is not None else ctx.empty((z.batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((z.batch, z.n), self.dtype)
        r = ctx.empty((z.batch, z.n), self.dtype) if want_posterior else None
        v = ctx.empty((z.batch,), np.float64) if want_posterior else None
        mask = ctx.empty((z.batch, z.n), np.uint8) if want_mask else None
        ctx._call("tramp_pwl_channel", self._dt, self._kind, direction, z.n, z.batch,
                  a_z.dev.ptr, z.dev.ptr, a_x.dev.ptr, x.dev.ptr, float(damping or 0.0),
                  old_a.ptr if old_a is not None else None, old_b.ptr if old_b is not None else None,
                  a_new.ptr, b_new.ptr, r.ptr if r is not None else None,
                  v.ptr if v is not None else None, mask.ptr if mask is not None else None)
        tgt, a_t = (x, a_x) if direction == _lib.FWD else (z, a_z)
        return tgt, a_t, a_new, b_new, r, v, mask

    def compute_log_partition(self, az, bz, ax, bx):
        "logsumexp of the regions' log-partitions, summed (piecewise_linear_channel.py:55-58)"
        kinds = {_lib.SGN: _lib.LOGZ_PWL_SGN, _lib.ABS: _lib.LOGZ_PWL_ABS, _lib.RELU: _lib.LOGZ_PWL_RELU}
        kind = kinds.get(getattr(self, "_kind", None)) if type(self) in (SgnChannel, AbsChannel, ReluChannel) else None
        if kind is None:
            # the rest of the family: the reference's loop over its region table (linear_region.py:55-64)
            return self._logz_regions(0, self._regions(), az, bz, bx, ax=ax, mean=False)
        return self._logz(kind, az, bz, a1=ax, v1=bx, mean=False)

    def compute_forward_posterior(self, az, bz, ax, bx):
        t, a, _, _, r, v, _ = self._run(_lib.FWD, az, bz, ax, bx, want_posterior=True)
        return ret_vec(r, t), ret_scal(v, a, t)

    def compute_backward_posterior(self, az, bz, ax, bx):
        t, a, _, _, r, v, _ = self._run(_lib.BWD, az, bz, ax, bx, want_posterior=True)
        return ret_vec(r, t), ret_scal(v, a, t)

    def compute_forward_message(self, az, bz, ax, bx, **kw):
        t, a, a_new, b_new, _, _, _ = self._run(_lib.FWD, az, bz, ax, bx, **kw)
        return ret_scal(a_new, a, t), ret_vec(b_new, t)

    def compute_backward_message(self, az, bz, ax, bx, **kw):
        t, a, a_new, b_new, _, _, _ = self._run(_lib.BWD, az, bz, ax, bx, **kw)
        return ret_scal(a_new, a, t), ret_vec(b_new, t)

    def branch_masks(self, az, bz, ax, bx):
        """log_Phi 'x < 30' masks of the (pos, neg) regions as two boolean arrays
        (tramp/utils/truncated_normal.py:27) - the index masks pinned bit-exact."""
        t, _, _, _, _, _, mask = self._run(_lib.BWD, az, bz, ax, bx, want_mask=True)
        m = mask.get().reshape(t.host_shape)
        return (m & 1).astype(bool), ((m >> 1) & 1).astype(bool)

    def _se(self, direction, az, ax, tau_z):
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        ax_h = np.broadcast_to(np.asarray(ax, dtype=np.float64), az_h.shape).copy()
        tz_h = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_pwl_channel", self._kind, direction, az_h.shape[0], az_h.ctypes.data,
                       ax_h.ctypes.data, tz_h.ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(az) == 0 else v

    def compute_forward_error(self, az, ax, tau_z):
        "channels/base_channel.py:29-34 with LinearRegion.beliefs_measure (K14)"
        return self._se(_lib.FWD, az, ax, tau_z)

    def compute_backward_error(self, az, ax, tau_z):
        "channels/base_channel.py:36-41 (K14)"
        return self._se(_lib.BWD, az, ax, tau_z)


class SgnChannel(PiecewiseLinearChannel):
    _kind = _lib.SGN
    _f = staticmethod(np.sign)


class AbsChannel(PiecewiseLinearChannel):
    _kind = _lib.ABS
    _f = staticmethod(np.abs)


class ReluChannel(PiecewiseLinearChannel):
    _kind = _lib.RELU
    _f = staticmethod(lambda z: np.maximum(0, z))


class TwoHalfLineChannel(PiecewiseLinearChannel):
    """Generic K7: z in [zcut, inf) -> x0_pos + slope_pos z, z in (-inf, zcut] -> x0_neg + slope_neg z
    (the remaining two-region members of the piecewise-linear family)."""
    _spec = (1.0, 1.0, 0.0, 0.0, 0.0)     # slope_pos, slope_neg, x0_pos, x0_neg, zcut

    def __init__(self, dtype=np.float64, ctx=None):
        self.repr_init()
        self._init_device(dtype, ctx)

    def sample(self, Z):
        sp, sn, xp, xn, zc = self._spec
        return np.where(Z >= zc, xp + sp * Z, xn + sn * Z)

    def _regions(self):
        "[pos, neg] rows (zmin, zmax, slope, x0) as the reference's constructors list them"
        sp, sn, xp, xn, zc = self._spec
        return [(zc, np.inf, sp, xp), (-np.inf, zc, sn, xn)]

    def second_moment(self, tau_z):
        "piecewise_linear_channel.py:21-25: sum_k p_k (rx_k^2 + vx_k) for z ~ N(0, tau_z)"
        from math import erfc, exp, pi, sqrt
        sp, sn, xp, xn, zc = self._spec
        s0 = sqrt(tau_z)
        y = zc / s0
        pdf = exp(-0.5 * y * y) / sqrt(2 * pi)
        p_pos = 0.5 * erfc(y / sqrt(2))
        p_neg = 1.0 - p_pos
        tot = 0.0
        for p, sgn, slope, x0 in ((p_pos, +1, sp, xp), (p_neg, -1, sn, xn)):
            if p <= 0:
                continue
            lam = sgn * pdf / p                       # E[z]/s0 of the truncated standard normal
            mean = s0 * lam
            var = tau_z * (1 + y * lam - lam * lam)
            tot += p * ((slope * mean + x0)**2 + slope * slope * var)
        return tot

    def _run(self, direction, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
             old_b=None, want_posterior=False, want_mask=False):
        ctx = self.ctx
        z = VecArg(ctx, bz, self.dtype)
        x = VecArg(ctx, bx, self.dtype)
        if (z.batch, z.n) != (x.batch, x.n):
            raise ValueError(f"bz {z.dev.shape} and bx {x.dev.shape} must have the same shape")
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        a_new = out_a if out_a is not None else ctx.empty((z.batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((z.batch, z.n), self.dtype)
        r = ctx.empty((z.batch, z.n), self.dtype) if want_posterior else None
        v = ctx.empty((z.batch,), np.float64) if want_posterior else None
        mask = ctx.empty((z.batch, z.n), np.uint8) if want_mask else None
        ctx._call("tramp_pwl2_channel", self._dt, direction, z.n, z.batch, *[float(p) for p in self._spec],
                  a_z.dev.ptr, z.dev.ptr, a_x.dev.ptr, x.dev.ptr, float(damping or 0.0),
                  old_a.ptr if old_a is not None else None, old_b.ptr if old_b is not None else None,
                  a_new.ptr, b_new.ptr, r.ptr if r is not None else None,
                  v.ptr if v is not None else None, mask.ptr if mask is not None else None)
        tgt, a_t = (x, a_x) if direction == _lib.FWD else (z, a_z)
        return tgt, a_t, a_new, b_new, r, v, mask

    def _se(self, direction, az, ax, tau_z):
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        ax_h = np.broadcast_to(np.asarray(ax, dtype=np.float64), az_h.shape).copy()
        tz_h = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_pwl2_channel", direction, az_h.shape[0], *[float(p) for p in self._spec],
                       az_h.ctypes.data, ax_h.ctypes.data, tz_h.ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(az) == 0 else v


class LeakyReluChannel(TwoHalfLineChannel):
    "tramp/channels/activation/piecewise_linear_channel.py:72-77"

    def __init__(self, slope, dtype=np.float64, ctx=None):
        self.slope = slope
        self.name = "l-relu"
        self._spec = (1.0, float(slope), 0.0, 0.0, 0.0)
        super().__init__(dtype, ctx)


class AsymmetricAbsChannel(TwoHalfLineChannel):
    "tramp/channels/activation/piecewise_linear_channel.py:94-99"

    def __init__(self, shift=1e-4, dtype=np.float64, ctx=None):
        self.shift = shift
        self.name = "a-abs"
        self._spec = (1.0, -1.0, 0.0, 0.0, float(shift))
        super().__init__(dtype, ctx)


class ThreeRegionChannel(PiecewiseLinearChannel):
    """Generic three-region K7: pos [z_hi, inf), mid [z_lo, z_hi], neg (-inf, z_lo] (SURVEY section 8f
    rank 2); the mid region is the finite-interval truncated normal of
    tramp/utils/truncated_normal.py:32-200 with its close/neg/pos/other branch masks."""
    _z = (-1.0, 1.0)
    _slopes = (0.0, 1.0, 0.0)     # pos, mid, neg
    _x0 = (1.0, 0.0, -1.0)

    def __init__(self, dtype=np.float64, ctx=None):
        self.repr_init()
        self._init_device(dtype, ctx)
        self._slopes_h = np.array(self._slopes, dtype=np.float64)
        self._x0_h = np.array(self._x0, dtype=np.float64)

    def sample(self, Z):
        zlo, zhi = self._z
        (sp, sm, sn), (xp, xm, xn) = self._slopes, self._x0
        return np.where(Z >= zhi, xp + sp * Z, np.where(Z >= zlo, xm + sm * Z, xn + sn * Z))

    def _regions(self):
        "[pos, mid, neg] rows (zmin, zmax, slope, x0)"
        zlo, zhi = self._z
        (sp, sm, sn), (xp, xm, xn) = self._slopes, self._x0
        return [(zhi, np.inf, sp, xp), (zlo, zhi, sm, xm), (-np.inf, zlo, sn, xn)]

    def second_moment(self, tau_z):
        "piecewise_linear_channel.py:21-25 with the truncated moments of N(0, tau_z) per region"
        from math import erf, exp, pi, sqrt
        s0 = sqrt(tau_z)
        Phi = lambda t: 0.5 * (1 + erf(t / sqrt(2)))     # noqa: E731
        phi = lambda t: exp(-0.5 * t * t) / sqrt(2 * pi)  # noqa: E731
        zlo, zhi = self._z
        tot = 0.0
        for (lo, hi), slope, x0 in zip(((zhi, None), (zlo, zhi), (None, zlo)), self._slopes, self._x0):
            a = -np.inf if lo is None else lo / s0
            b = np.inf if hi is None else hi / s0
            Pa, Pb = (0.0 if lo is None else Phi(a)), (1.0 if hi is None else Phi(b))
            pa, pb = (0.0 if lo is None else phi(a)), (0.0 if hi is None else phi(b))
            Z = Pb - Pa
            if Z <= 0:
                continue
            lam = (pa - pb) / Z
            apa = 0.0 if lo is None else a * pa
            bpb = 0.0 if hi is None else b * pb
            mean = s0 * lam
            var = tau_z * (1 + (apa - bpb) / Z - lam * lam)
            tot += Z * ((slope * mean + x0)**2 + slope * slope * var)
        return tot

    def _run(self, direction, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
             old_b=None, want_posterior=False, want_mask=False):
        ctx = self.ctx
        z = VecArg(ctx, bz, self.dtype)
        x = VecArg(ctx, bx, self.dtype)
        if (z.batch, z.n) != (x.batch, x.n):
            raise ValueError(f"bz {z.dev.shape} and bx {x.dev.shape} must have the same shape")
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        a_new = out_a if out_a is not None else ctx.empty((z.batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((z.batch, z.n), self.dtype)
        r = ctx.empty((z.batch, z.n), self.dtype) if want_posterior else None
        v = ctx.empty((z.batch,), np.float64) if want_posterior else None
        mask = ctx.empty((z.batch, z.n), np.uint8) if want_mask else None
        ctx._call("tramp_pwl3_channel", self._dt, direction, z.n, z.batch, float(self._z[0]), float(self._z[1]),
                  self._slopes_h.ctypes.data, self._x0_h.ctypes.data,
                  a_z.dev.ptr, z.dev.ptr, a_x.dev.ptr, x.dev.ptr, float(damping or 0.0),
                  old_a.ptr if old_a is not None else None, old_b.ptr if old_b is not None else None,
                  a_new.ptr, b_new.ptr, r.ptr if r is not None else None,
                  v.ptr if v is not None else None, mask.ptr if mask is not None else None)
        tgt, a_t = (x, a_x) if direction == _lib.FWD else (z, a_z)
        return tgt, a_t, a_new, b_new, r, v, mask

    def branch_masks(self, az, bz, ax, bx):
        """(close, neg, pos, other) boolean masks of the mid region's finite-interval branch
        (tramp/utils/truncated_normal.py:68-82) - index masks, pinned bit-exact in f64."""
        t, _, _, _, _, _, mask = self._run(_lib.BWD, az, bz, ax, bx, want_mask=True)
        code = (mask.get().reshape(t.host_shape) >> 2) & 3
        return tuple(code == k for k in range(4))

    def _se(self, direction, az, ax, tau_z):
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        ax_h = np.broadcast_to(np.asarray(ax, dtype=np.float64), az_h.shape).copy()
        tz_h = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_pwl3_channel", direction, az_h.shape[0], float(self._z[0]), float(self._z[1]),
                       self._slopes_h.ctypes.data, self._x0_h.ctypes.data,
                       az_h.ctypes.data, ax_h.ctypes.data, tz_h.ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(az) == 0 else v


class HardTanhChannel(ThreeRegionChannel):
    "tramp/channels/activation/piecewise_linear_channel.py:109-114"
    name = "h-tanh"


class HardSigmoidChannel(ThreeRegionChannel):
    "piecewise_linear_channel.py:117-123 (L = 2.5)"
    name = "h-sigm"
    _z = (-2.5, 2.5)
    _slopes = (0.0, 1 / 5.0, 0.0)
    _x0 = (1.0, 0.5, 0.0)


class SymmetricDoorChannel(ThreeRegionChannel):
    "piecewise_linear_channel.py:126-132"
    name = "door"

    def __init__(self, width, dtype=np.float64, ctx=None):
        self.width = width
        self._z = (-float(width), float(width))
        self._slopes = (0.0, 0.0, 0.0)
        self._x0 = (1.0, -1.0, 1.0)
        super().__init__(dtype, ctx)


# ------------------------------------------------------------------ dense linear
def thin_svd_host(W):
    """Thin SVD on the host (LAPACK), the factorisation LinearChannel.__init__ computes
    (tramp/channels/linear/linear_channel.py:8-15,37-41) minus the parts that only
    multiply zeros.  Returns Uk [M,k], sing [k], Vt [k,N], rank."""
    W = np.asarray(W, dtype=np.float64)
    U, s, Vt = np.linalg.svd(W, full_matrices=False)
    tol = s.max() * max(W.shape) * np.finfo(np.float64).eps     # np.linalg.matrix_rank
    rank = int(np.sum(s > tol))
    return np.ascontiguousarray(U), s, np.ascontiguousarray(Vt), rank


class LinearChannel(Channel):
    """Linear channel x = W z (kernels K8, K9).

    tramp/channels/linear/linear_channel.py:18-143.  ``W`` may be a NumPy array
    (uploaded) or a DeviceArray [M, N].  ``svd``: "host" (LAPACK), "device"
    (tramp_linear_factorize: rocBLAS Gram + rocSOLVER eigensolve) or "auto"
    (host below 2048 on the smaller side).  Pre-computed thin factors can be
    passed as ``factors=(Uk, sing, Vt, rank)``.  W itself is not kept on the
    device: only Uk [M,k], Vt [k,N], sing enter the posterior (see csrc/linear.hip).
    """

    def __new__(cls, W=None, precompute_svd=True, *args, **kwargs):
        # precompute_svd=False is a different device object (csrc/solve.hip); still a LinearChannel for isinstance
        if cls is LinearChannel and not precompute_svd:
            return super().__new__(SolveLinearChannel)
        return super().__new__(cls)

    def __init__(self, W, precompute_svd=True, name="W", dtype=np.float64, ctx=None, svd="auto",
                 factors=None, keep_W=False, digits=None):
        self.name = name
        self.Nx, self.Nz = (int(s) for s in W.shape)
        self.precompute_svd = precompute_svd
        self.repr_init()
        self._init_device(dtype, ctx)
        self.alpha = self.Nx / self.Nz
        ctx = self.ctx
        k = min(self.Nx, self.Nz)
        if factors is None:
            on_dev = isinstance(W, DeviceArray)
            if svd == "auto":
                svd = "device" if (on_dev or k > 2048) else "host"
            if svd == "host":
                Wh = W.get() if on_dev else W
                factors = thin_svd_host(Wh)
            else:
                Wd = W if on_dev else ctx.array(np.asarray(W, dtype=self.dtype), self.dtype)
                if Wd.dtype != self.dtype:
                    Wd = Wd.astype(self.dtype)
                Uk = ctx.empty((self.Nx, k), self.dtype)
                Vt = ctx.empty((k, self.Nz), self.dtype)
                sing = ctx.empty((k,), np.float64)
                rank = C.c_size_t()
                ctx._call("tramp_linear_factorize", self._dt, self.Nx, self.Nz, Wd.ptr, Uk.ptr, Vt.ptr,
                          sing.ptr, C.byref(rank))
                factors = (Uk, sing, Vt, rank.value)
        Uk, sing, Vt, rank = factors
        self._Uk = Uk if isinstance(Uk, DeviceArray) else ctx.array(Uk, self.dtype)
        self._Vt = Vt if isinstance(Vt, DeviceArray) else ctx.array(Vt, self.dtype)
        self._sing = sing if isinstance(sing, DeviceArray) else ctx.array(sing, np.float64)
        self.rank = int(rank)
        sing_h = self._sing.get()
        self.spectrum = np.zeros(self.Nz)                       # diag(S^T S), linear_channel.py:41
        self.spectrum[:k] = sing_h**2
        self.singular = self.spectrum[:self.rank]
        # a host matrix stays referenced (it costs no device memory); a device matrix only on request - the 40 GB W of the
        # headline run is freed by its caller once the factors exist.  sample() works either way.
        self.W = W if (keep_W or not isinstance(W, DeviceArray)) else None
        h = C.c_void_p()
        ctx._call("tramp_linear_create", self._dt, self.Nx, self.Nz, self.rank, self._Uk.ptr,
                  self._Vt.ptr, self._sing.ptr, C.byref(h))
        self._handle = h
        self._t_key = self._u_key = None
        # opt-in: the dense passes of ensembles on the int8 matrix cores (csrc/gemm_sliced.h).  digits=None reads the
        # switch TRAMP_MATVEC_DIGITS (0 = off; 7 is f64-equivalent, 4 is the f32 setting).
        # round 5: "auto" (the default of the switch) leaves the decision to the chain scheduler, which attaches them when an
        # ENSEMBLE of >= 3 instances iterates and device memory allows (algos.py: _auto_attach_digits) - 1.4x on the dense
        # passes at f64 accuracy, parity-tested against the goldens like the matrix-core engines
        # (tests/test_gpu_digits_engine_suite.py); digits=0 / TRAMP_MATVEC_DIGITS=0 keeps the f64 / f32 matrix cores.
        self._digits_auto = False
        if digits is None:
            env = os.environ.get("TRAMP_MATVEC_DIGITS", "auto").strip().lower()
            if env in ("auto", ""):
                digits, self._digits_auto = 0, True
            else:
                digits = int(env)
        self.digits = 0
        if digits:
            self.attach_digits(int(digits))

    def attach_digits(self, digits=0, uk=True):
        """Digit copies of the factors: Vt in both forms and (uk=True) Uk likewise - ~2 x digits / itemsize the memory of
        the factors on top of them.  uk=False when the scheduler carries the output side in the left singular basis
        (Uk then only rotates y once).  digits=0: by dtype (7 / 4)."""
        self.ctx.attach_digits(self._Vt, digits, 3)
        if uk:
            self.ctx.attach_digits(self._Uk, digits, 3)
        self.digits = self.ctx.digits_info(self._Vt)[0]

    def detach_digits(self):
        self.ctx.detach_digits(self._Vt)
        self.ctx.detach_digits(self._Uk)
        self.digits = 0

    def __del__(self):
        try:
            if getattr(self, "digits", 0):
                self.detach_digits()
        except Exception:
            pass
        try:
            self.ctx.release("tramp_linear_destroy", self._handle)
        except Exception:
            pass
        self._handle = None

    def sample(self, Z):
        "x = W z (linear_channel.py:48-50); without a kept W through the factors, W z = Uk (s * (Vt z)), two dense passes on the device"
        if self.W is not None:
            W = self.W.get() if isinstance(self.W, DeviceArray) else self.W
            return W @ Z
        ctx = self.ctx
        Zh = np.asarray(Z, dtype=self.dtype)
        z2 = Zh.reshape(1, -1) if Zh.ndim == 1 else Zh.reshape(Zh.shape[0], -1)
        k = self._Vt.shape[0]
        zd, t = ctx.array(np.ascontiguousarray(z2), self.dtype), ctx.empty((z2.shape[0], k), self.dtype)
        ctx._call("tramp_matvec_nt", self._dt, 0, k, self.Nz, z2.shape[0], self._Vt.ptr, zd.ptr, t.ptr)
        th = t.get() * self._sing.get()[None, :].astype(self.dtype)
        x = ctx.empty((z2.shape[0], self.Nx), self.dtype)
        ctx._call("tramp_matvec_nt", self._dt, 0, self.Nx, k, z2.shape[0], self._Uk.ptr, ctx.array(th, self.dtype).ptr, x.ptr)
        xh = x.get().astype(np.float64)
        return xh[0] if Zh.ndim == 1 else xh

    def second_moment(self, tau_z):
        return tau_z * self.spectrum.sum() / self.Nx            # linear_channel.py:55-56

    # -- scalars (K9); host closed forms for host scalars, kernel for device scalars
    def compute_n_eff(self, az, ax):
        "linear_channel.py:58-67 (kernel K9, tramp_linear_variance; the forward query leaves az unfloored)"
        return float(self.device_variance(_lib.FWD, az, ax)[2][0])

    def compute_backward_variance(self, az, ax):
        "linear_channel.py:91-97 (K9; the kernel floors az at 1e-11 like the reference)"
        return float(self.device_variance(_lib.BWD, az, ax)[0][0])

    def compute_forward_variance(self, az, ax):
        "linear_channel.py:99-105 (K9)"
        return float(self.device_variance(_lib.FWD, az, ax)[0][0])

    def compute_backward_error(self, az, ax, tau_z):
        return self.compute_backward_variance(az, ax)           # :119-121

    def compute_forward_error(self, az, ax, tau_z):
        return self.compute_forward_variance(az, ax)            # :123-125

    def device_variance(self, direction, az, ax):
        "K9 on the device for a batch of (az, ax); returns (v, a_new, n_eff) as NumPy arrays."
        ctx = self.ctx
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        a_z = ctx.scalars(az_h)
        a_x = ctx.scalars(ax, az_h.shape[0])
        outs = [ctx.empty((az_h.shape[0],), np.float64) for _ in range(3)]
        ctx._call("tramp_linear_variance", self._handle, direction, az_h.shape[0], a_z.ptr, a_x.ptr,
                  *[o.ptr for o in outs])
        return [o.get() for o in outs]

    # -- means + message (K8)
    def _run(self, direction, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
             old_b=None, want_posterior=False):
        ctx = self.ctx
        z = VecArg(ctx, bz, self.dtype)
        x = VecArg(ctx, bx, self.dtype)
        # input side carried in the right singular basis by the scheduler (tramp_linear_set_rotated bit 1)
        nz = min(self.Nx, self.Nz) if getattr(self, "_rot_mask", 0) & 2 else self.Nz
        if z.n != nz or x.n != self.Nx or z.batch != x.batch:
            raise ValueError(f"expected bz [batch,{nz}] and bx [batch,{self.Nx}], "
                             f"got {z.dev.shape} and {x.dev.shape}")
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        tgt, a_t = (x, a_x) if direction == _lib.FWD else (z, a_z)
        a_new = out_a if out_a is not None else ctx.empty((z.batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((tgt.batch, tgt.n), self.dtype)
        r = ctx.empty((tgt.batch, tgt.n), self.dtype) if want_posterior else None
        v = ctx.empty((z.batch,), np.float64) if want_posterior else None
        # projection cache: t = Vt bz and u = Uk^T bx persist in the handle; a device-resident
        # message that has not been rewritten since (same buffer, same version) needs no new pass
        reuse = 0
        kz, kx = self._proj_key(bz, z), self._proj_key(bx, x)
        if kz is not None and kz == self._t_key:
            reuse |= _lib.REUSE_T
        if kx is not None and kx == self._u_key:
            reuse |= _lib.REUSE_U
        self._t_key, self._u_key = kz, kx
        ctx._call("tramp_linear_message", self._handle, direction, z.batch, a_z.dev.ptr, z.dev.ptr,
                  a_x.dev.ptr, x.dev.ptr, float(damping or 0.0),
                  old_a.ptr if old_a is not None else None, old_b.ptr if old_b is not None else None,
                  a_new.ptr, b_new.ptr, r.ptr if r is not None else None,
                  v.ptr if v is not None else None, reuse)
        return tgt, a_t, a_new, b_new, r, v

    @staticmethod
    def _proj_key(arg, norm):
        "Identity of a device-resident message for the projection cache (None: not cacheable)."
        version = getattr(arg, "version", None)
        if not isinstance(arg, DeviceArray) or version is None:
            return None
        return (arg.ptr, version, norm.batch)

    def compute_log_partition(self, az, bz, ax, bx):
        "linear_channel.py:127-132, from the thin factors (the projections are shared with the messages)"
        ctx = self.ctx
        z = VecArg(ctx, bz, self.dtype)
        x = VecArg(ctx, bx, self.dtype)
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        reuse = 0
        kz, kx = self._proj_key(bz, z), self._proj_key(bx, x)
        if kz is not None and kz == self._t_key:
            reuse |= _lib.REUSE_T
        if kx is not None and kx == self._u_key:
            reuse |= _lib.REUSE_U
        self._t_key, self._u_key = kz, kx
        out = ctx.empty((z.batch,), np.float64)
        ctx._call("tramp_linear_log_partition", self._handle, z.batch, a_z.dev.ptr, z.dev.ptr, a_x.dev.ptr,
                  x.dev.ptr, reuse, out.ptr)
        res = out.get()
        return float(res[0]) if (a_z.was_host and a_z.scalar_like and z.was_host and len(z.host_shape) == 1) else res

    _affine_entry = "tramp_linear_project_affine"

    def affine_update(self, side, old_key, msg, c, damping, token=None):
        """``msg`` (the channel's bz for side "z", bx for "x") was just replaced in place by
        damping * msg + (1 - damping) * c with a constant c (constant damping of a constant
        factor's message).  If the cached projection belongs to the message as it was before
        (``old_key``), it follows by linearity - O(k) instead of a matrix pass - and stays valid."""
        attr = "_t_key" if side == "z" else "_u_key"
        if old_key is None or getattr(self, attr) != old_key:
            return False
        tokens = self.__dict__.setdefault("_c_tokens", {})
        changed = tokens.get(side) != token or token is None
        self.ctx._call(self._affine_entry, self._handle, _lib.REUSE_T if side == "z" else _lib.REUSE_U,
                       old_key[2], c.ptr, float(damping), int(changed))
        tokens[side] = token
        setattr(self, attr, (msg.ptr, msg.version, old_key[2]))
        return True

    def compute_backward_mean(self, az, bz, ax, bx):
        "linear_channel.py:69-83"
        return self.compute_backward_posterior(az, bz, ax, bx)[0]

    def compute_forward_mean(self, az, bz, ax, bx):
        "linear_channel.py:85-89"
        return self.compute_forward_posterior(az, bz, ax, bx)[0]

    def compute_backward_posterior(self, az, bz, ax, bx):
        t, a, _, _, r, v = self._run(_lib.BWD, az, bz, ax, bx, want_posterior=True)
        return ret_vec(r, t), ret_scal(v, a, t)

    def compute_forward_posterior(self, az, bz, ax, bx):
        t, a, _, _, r, v = self._run(_lib.FWD, az, bz, ax, bx, want_posterior=True)
        return ret_vec(r, t), ret_scal(v, a, t)

    def compute_forward_message(self, az, bz, ax, bx, **kw):
        t, a, a_new, b_new, _, _ = self._run(_lib.FWD, az, bz, ax, bx, **kw)
        return ret_scal(a_new, a, t), ret_vec(b_new, t)

    def compute_backward_message(self, az, bz, ax, bx, **kw):
        t, a, a_new, b_new, _, _ = self._run(_lib.BWD, az, bz, ax, bx, **kw)
        return ret_scal(a_new, a, t), ret_vec(b_new, t)


class ComplexLinearChannel(Channel):
    """Complex linear channel x = W z, W complex [Nx, Nz]; messages are planar real arrays (2, N).

    tramp/channels/linear/complex_linear_channel.py:17-162.  On the device the channel IS the dense LinearChannel of the
    real embedding [[Re W, -Im W], [Im W, Re W]] acting on [Re z; Im z] (the planar layout flattened): every singular
    value of W appears twice, so n_eff, both variances, alpha and the log-partition are those of the complex channel
    (pinned on the reference's outputs in tests/golden/complex.npz) - the same MFMA / GEMV kernels, the same
    factorisation.  Host messages (2, n) or (batch, 2, n); device messages [batch, 2 n].
    """

    def __init__(self, W, precompute_svd=True, name="W", dtype=np.float64, ctx=None, svd="auto", keep_W=False):
        if not precompute_svd:
            raise NotImplementedError("ComplexLinearChannel(precompute_svd=False) is outside the HIP path")
        W = np.asarray(W)
        self.name = name
        self.Nx, self.Nz = (int(s) for s in W.shape)
        self.precompute_svd = precompute_svd
        self.repr_init()
        self._init_device(dtype, ctx)
        self.alpha = self.Nx / self.Nz
        Wr = np.block([[W.real, -W.imag], [W.imag, W.real]])
        self._real = LinearChannel(Wr, name=name, dtype=dtype, ctx=self.ctx, svd=svd, digits=0)
        self.rank = self._real.rank // 2
        sv = np.sqrt(self._real.spectrum)                     # every singular value of W twice, descending
        self.spectrum = (sv[0::2] ** 2)[:self.Nz]              # complex_linear_channel.py:50-56
        self.singular = self.spectrum[:self.rank]
        self.W = W if keep_W else None

    # the scheduler talks to the embedded channel's device state
    @property
    def _ctx(self):
        return self.__dict__.get("_ctx_")

    @_ctx.setter
    def _ctx(self, c):
        self.__dict__["_ctx_"] = c
        if "_real" in self.__dict__ and c is not None:
            self._real._ctx = c

    def affine_update(self, *a, **k):
        return self._real.affine_update(*a, **k)

    def sample(self, Z):
        if self.W is None:
            raise RuntimeError("ComplexLinearChannel was built with keep_W=False")
        Z = np.asarray(Z)
        X = self.W.dot(Z[0] + 1j * Z[1])
        return np.stack([X.real, X.imag])

    def second_moment(self, tau_z):
        return tau_z * self.spectrum.sum() / self.Nx            # :72-73

    def compute_n_eff(self, az, ax):
        return self._real.compute_n_eff(az, ax)

    def compute_backward_variance(self, az, ax):
        assert az > 0                                            # :109
        return self._real.compute_backward_variance(az, ax)

    def compute_forward_variance(self, az, ax):
        return self._real.compute_forward_variance(az, ax)

    def compute_backward_error(self, az, ax, tau_z):
        return self.compute_backward_variance(az, ax)

    def compute_forward_error(self, az, ax, tau_z):
        return self.compute_forward_variance(az, ax)

    @staticmethod
    def _flat(b):
        "planar host array (2, n) / (batch, 2, n) -> (2 n,) / (batch, 2 n); device arrays pass through"
        if isinstance(b, DeviceArray):
            return b, None
        h = np.asarray(b)
        if h.ndim not in (2, 3) or h.shape[-2] != 2:
            raise ValueError(f"expected a planar complex message (2, n) or (batch, 2, n), got {h.shape}")
        return h.reshape(h.shape[:-2] + (2 * h.shape[-1],)), h.shape[:-2]

    def _call(self, method, az, bz, ax, bx, **kw):
        fz, lead = self._flat(bz)
        fx, _ = self._flat(bx)
        s, vec = getattr(self._real, method)(az, fz, ax, fx, **kw)
        if method.endswith("posterior"):
            vec, s = s, vec
        if lead is not None and not isinstance(vec, DeviceArray):
            vec = vec.reshape(lead + (2, vec.shape[-1] // 2))
        return (vec, s) if method.endswith("posterior") else (s, vec)

    def compute_backward_posterior(self, az, bz, ax, bx):
        return self._call("compute_backward_posterior", az, bz, ax, bx)

    def compute_forward_posterior(self, az, bz, ax, bx):
        return self._call("compute_forward_posterior", az, bz, ax, bx)

    def compute_backward_mean(self, az, bz, ax, bx):
        return self.compute_backward_posterior(az, bz, ax, bx)[0]

    def compute_forward_mean(self, az, bz, ax, bx):
        return self.compute_forward_posterior(az, bz, ax, bx)[0]

    def compute_backward_message(self, az, bz, ax, bx, **kw):
        return self._call("compute_backward_message", az, bz, ax, bx, **kw)

    def compute_forward_message(self, az, bz, ax, bx, **kw):
        return self._call("compute_forward_message", az, bz, ax, bx, **kw)

    def compute_log_partition(self, az, bz, ax, bx):
        "complex_linear_channel.py:143-150 = the real embedding's 0.5 sum(b rz) + 0.5 sum log(2 pi / a) over 2 Nz entries"
        return self._real.compute_log_partition(az, self._flat(bz)[0], ax, self._flat(bx)[0])


class SolveLinearChannel(LinearChannel):
    """LinearChannel(W, precompute_svd=False): the reference's branch that keeps C = W^T W and solves
    (az I + ax C) rz = bz + W^T bx on every call (tramp/channels/linear/linear_channel.py:43-46, 79-82) - on the
    device one Cholesky factorisation per instance and call (csrc/solve.hip: rocSOLVER potrf / potrs), the matrix
    products on W itself by the channel's own matrix kernels.  Built by ``LinearChannel(W, precompute_svd=False)``.

    As in the reference ``spectrum = eigvalsh(C)`` is ASCENDING here and ``singular = spectrum[:rank]`` keeps its
    first - smallest - ``rank`` entries; n_eff and the variances follow that array, so for Nz > rank they differ
    from the precompute_svd=True channel exactly as the reference's two branches differ.  ``rank`` is
    np.linalg.matrix_rank(W) for a host array; for a device-resident W it is counted from the spectrum
    (sigma > sigma_max * max(M, N) * eps, resolved only down to ~1e-8 sigma_max by the Gram form).
    The scheduler never rotates this channel into a singular basis and cannot capture it in a hipGraph."""

    def attach_digits(self, digits=0, uk=True):
        raise ValueError("the int8-digit engine serves the factorised channel (precompute_svd=True) only")

    def __init__(self, W, precompute_svd=False, name="W", dtype=np.float64, ctx=None, svd="auto", factors=None,
                 keep_W=True, digits=None):
        if precompute_svd:
            raise ValueError("SolveLinearChannel is the precompute_svd=False branch")
        if digits:
            raise ValueError("digits: the int8-digit engine serves the factorised channel (precompute_svd=True) only")
        self.name = name
        self.Nx, self.Nz = (int(s) for s in W.shape)
        self.precompute_svd = False
        self.repr_init()
        self._init_device(dtype, ctx)
        self.alpha = self.Nx / self.Nz
        ctx = self.ctx
        on_dev = isinstance(W, DeviceArray)
        self._Wd = W if on_dev and W.dtype == self.dtype else (
            W.astype(self.dtype) if on_dev else ctx.array(np.asarray(W, dtype=self.dtype), self.dtype))
        self.W = W
        h = C.c_void_p()
        ctx._call("tramp_lsolve_create", self._dt, self.Nx, self.Nz, self._Wd.ptr, C.byref(h))
        self._handle = h
        spec = (C.c_double * self.Nz)()
        ctx._call("tramp_lsolve_spectrum", self._handle, spec)
        self.spectrum = np.frombuffer(spec, dtype=np.float64).copy()            # ascending (:44)
        if on_dev:
            s = np.sqrt(np.maximum(self.spectrum, 0.0))
            self.rank = int(np.sum(s > s.max() * max(self.Nx, self.Nz) * np.finfo(np.float64).eps))
        else:
            self.rank = int(np.linalg.matrix_rank(np.asarray(W, dtype=np.float64)))   # :37
        ctx._call("tramp_lsolve_set_rank", self._handle, self.rank)
        self.singular = self.spectrum[:self.rank]                               # :46
        self._t_key = self._u_key = None

    def __del__(self):
        try:
            self.ctx.release("tramp_lsolve_destroy", self._handle)
        except Exception:
            pass
        self._handle = None

    def sample(self, Z):
        W = self.W.get() if isinstance(self.W, DeviceArray) else self.W
        return W @ Z

    def device_variance(self, direction, az, ax):
        "(v, a_new, n_eff) from the reference's closed forms over `singular` (linear_channel.py:58-67, 91-105)"
        az = np.atleast_1d(np.asarray(az, dtype=np.float64))
        ax = np.broadcast_to(np.asarray(ax, dtype=np.float64), az.shape)
        v, a_new, n_eff = np.empty_like(az), np.empty_like(az), np.empty_like(az)
        for i, (a_z, a_x) in enumerate(zip(az, ax)):
            azc = max(1e-11, a_z) if direction == _lib.BWD else a_z
            if a_x == 0:
                ne = 0.0
            elif azc / a_x == 0:
                ne = self.rank / self.Nz
            else:
                ne = float(np.sum(self.singular / (azc / a_x + self.singular))) / self.Nz
            if direction == _lib.BWD:
                vi, a_in = (1 - ne) / azc, a_z
            elif a_x == 0:
                vi, a_in = float(np.mean(self.singular)) * self.rank / (self.Nx * a_z), a_x
            else:
                vi, a_in = ne / (self.alpha * a_x), a_x
            v[i], n_eff[i] = vi, ne
            a_new[i] = self.compute_a_new(vi, a_in)
        return [v, a_new, n_eff]

    def _run(self, direction, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None,
             want_posterior=False):
        ctx = self.ctx
        z = VecArg(ctx, bz, self.dtype)
        x = VecArg(ctx, bx, self.dtype)
        if z.n != self.Nz or x.n != self.Nx or z.batch != x.batch:
            raise ValueError(f"expected bz [batch,{self.Nz}] and bx [batch,{self.Nx}], "
                             f"got {z.dev.shape} and {x.dev.shape}")
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        tgt, a_t = (x, a_x) if direction == _lib.FWD else (z, a_z)
        a_new = out_a if out_a is not None else ctx.empty((z.batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((tgt.batch, tgt.n), self.dtype)
        r = ctx.empty((tgt.batch, tgt.n), self.dtype) if want_posterior else None
        v = ctx.empty((z.batch,), np.float64) if want_posterior else None
        ctx._call("tramp_lsolve_message", self._handle, direction, z.batch, a_z.dev.ptr, z.dev.ptr, a_x.dev.ptr,
                  x.dev.ptr, float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr,
                  r.ptr if r is not None else None, v.ptr if v is not None else None)
        return tgt, a_t, a_new, b_new, r, v

    def compute_log_partition(self, az, bz, ax, bx):
        "linear_channel.py:127-132 with rz from the solve"
        ctx = self.ctx
        z = VecArg(ctx, bz, self.dtype)
        x = VecArg(ctx, bx, self.dtype)
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        out = ctx.empty((z.batch,), np.float64)
        ctx._call("tramp_lsolve_log_partition", self._handle, z.batch, a_z.dev.ptr, z.dev.ptr, a_x.dev.ptr, x.dev.ptr,
                  out.ptr)
        res = out.get()
        return float(res[0]) if (a_z.was_host and a_z.scalar_like and z.was_host and len(z.host_shape) == 1) else res

    def affine_update(self, side, old_key, msg, c, damping, token=None):
        return False                                                          # no cached projections to carry along


class ConvChannel(Channel):
    """Circular convolution channel x = w * z for real signals (1-3 axes).

    tramp/channels/linear/conv_channel.py:13-149 (``real=True``).  ``filter`` is the
    filter f; the convolution weights are w[u] = f[-u] (fft(w) = conj(fft(f))).  All
    O(n log n): batched rocFFT real transforms around hand-written spectral kernels
    (csrc/conv.hip); no dense matrix exists at any point, so n = 10^6 and beyond fit.
    Vectors are [batch, prod(shape)] on the device; host arrays of shape ``shape`` or
    ``(batch,) + shape`` are accepted and returned in kind.
    """

    def __init__(self, filter, real=True, name="W", dtype=np.float64, ctx=None):
        filter = np.asarray(filter, dtype=np.float64 if real else np.complex128)
        if not 1 <= filter.ndim <= 3:
            raise ValueError("filter must have 1 to 3 axes")
        self.shape = filter.shape
        self.real = bool(real)
        self.name = name
        self.repr_init()
        self._init_device(dtype, ctx)
        self.filter = filter
        n = int(np.prod(self.shape))
        # real=False (conv_channel.py:25-32): x, w, z complex; every vector is the planar real array (2,) + shape
        self._vshape = tuple(self.shape) if real else (2,) + tuple(self.shape)
        self.N = n if real else 2 * n                           # vector length per instance on the device
        self.Nx = self.Nz = self.N
        ctx = self.ctx
        dims = (C.c_size_t * filter.ndim)(*self.shape)
        h = C.c_void_p()
        if real:
            fd = ctx.array(filter.reshape(-1), np.float64)
            ctx._call("tramp_conv_create", self._dt, filter.ndim, dims, fd.ptr, C.byref(h))
        else:
            fd = ctx.array(np.concatenate([filter.real.reshape(-1), filter.imag.reshape(-1)]), np.float64)
            ctx._call("tramp_conv_create_complex", self._dt, filter.ndim, dims, fd.ptr, C.byref(h))
        self._handle = h
        cnt, mean = C.c_size_t(), C.c_double()
        ctx._call("tramp_conv_spectrum", h, None, C.byref(cnt), C.byref(mean))
        self._n_half = cnt.value
        self.spectrum_mean = mean.value
        self._t_key = self._u_key = None

    def __del__(self):
        try:
            self.ctx.release("tramp_conv_destroy", self._handle)
        except Exception:
            pass
        self._handle = None

    @property
    def half_spectrum(self):
        "|rfftn(filter)|^2 flattened (last axis n_last // 2 + 1 bins); real=False: the full |fftn(filter)|^2."
        if not self.real:
            return (np.abs(np.fft.fftn(self.filter))**2).reshape(-1)
        out = self.ctx.empty((self._n_half,), np.float64)
        self.ctx._call("tramp_conv_spectrum", self._handle, out.ptr, None, None)
        return out.get()

    # -- argument normalisation: flatten the signal axes
    def _vec(self, b):
        if isinstance(b, DeviceArray):
            return VecArg(self.ctx, b, self.dtype)
        h = np.asarray(b, dtype=self.dtype)
        nd = len(self._vshape)
        if h.shape[-nd:] != self._vshape or h.ndim not in (nd, nd + 1):
            raise ValueError(f"expected an array of shape {self._vshape} or (batch,)+{self._vshape}, got {h.shape}")
        arg = VecArg(self.ctx, h.reshape(-1, self.N) if h.ndim == nd + 1 else h.reshape(self.N), self.dtype)
        arg.host_shape = h.shape
        arg._one = h.ndim == nd
        return arg

    @staticmethod
    def _ret(dev, like):
        if like.was_host:
            return dev.get().reshape(like.host_shape)
        return ret_vec(dev, like)

    def convolve(self, z):
        "conv_channel.py:47-55"
        zz = self._vec(z)
        x = self.ctx.empty((zz.batch, self.N), self.dtype)
        self.ctx._call("tramp_conv_apply", self._handle, zz.batch, zz.dev.ptr, x.ptr)
        self._t_key = self._u_key = None                       # the call reuses the projection buffers
        return self._ret(x, zz)

    def sample(self, Z):
        return self.convolve(Z)                                # :57-63

    def second_moment(self, tau_z):
        return tau_z * self.spectrum_mean                      # :66-67

    def device_variance(self, direction, az, ax):
        "n_eff and the variances on the device for a batch of (az, ax): (v, a_new, n_eff)."
        ctx = self.ctx
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        a_z = ctx.scalars(az_h)
        a_x = ctx.scalars(ax, az_h.shape[0])
        outs = [ctx.empty((az_h.shape[0],), np.float64) for _ in range(3)]
        ctx._call("tramp_conv_variance", self._handle, direction, az_h.shape[0], a_z.ptr, a_x.ptr,
                  *[o.ptr for o in outs])
        return [o.get() for o in outs]

    def compute_n_eff(self, az, ax):
        return float(self.device_variance(_lib.BWD, az, ax)[2][0])     # :69-79

    def compute_backward_variance(self, az, ax):
        assert az > 0                                                  # :111-115
        return float(self.device_variance(_lib.BWD, az, ax)[0][0])

    def compute_forward_variance(self, az, ax):
        return float(self.device_variance(_lib.FWD, az, ax)[0][0])     # :117-123

    def compute_backward_error(self, az, ax, tau_z):
        return self.compute_backward_variance(az, ax)                  # :137-139

    def compute_forward_error(self, az, ax, tau_z):
        return self.compute_forward_variance(az, ax)                   # :141-143

    def _run(self, direction, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
             old_b=None, want_posterior=False):
        ctx = self.ctx
        z, x = self._vec(bz), self._vec(bx)
        if z.n != self.N or x.n != self.N or z.batch != x.batch:
            raise ValueError(f"expected bz and bx of [batch,{self.N}], got {z.dev.shape} and {x.dev.shape}")
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        tgt, a_t = (x, a_x) if direction == _lib.FWD else (z, a_z)
        a_new = out_a if out_a is not None else ctx.empty((z.batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((z.batch, self.N), self.dtype)
        r = ctx.empty((z.batch, self.N), self.dtype) if want_posterior else None
        v = ctx.empty((z.batch,), np.float64) if want_posterior else None
        reuse = 0
        kz, kx = LinearChannel._proj_key(bz, z), LinearChannel._proj_key(bx, x)
        if kz is not None and kz == self._t_key:
            reuse |= _lib.REUSE_T
        if kx is not None and kx == self._u_key:
            reuse |= _lib.REUSE_U
        self._t_key, self._u_key = kz, kx
        ctx._call("tramp_conv_message", self._handle, direction, z.batch, a_z.dev.ptr, z.dev.ptr,
                  a_x.dev.ptr, x.dev.ptr, float(damping or 0.0),
                  old_a.ptr if old_a is not None else None, old_b.ptr if old_b is not None else None,
                  a_new.ptr, b_new.ptr, r.ptr if r is not None else None,
                  v.ptr if v is not None else None, reuse)
        return tgt, a_t, a_new, b_new, r, v

    _affine_entry = "tramp_conv_project_affine"

    def affine_update(self, side, old_key, msg, c, damping, token=None):
        if not self.real:
            return False                                               # the complex channel keeps no separate constant part
        return LinearChannel.affine_update(self, side, old_key, msg, c, damping, token)

    def compute_backward_mean(self, az, bz, ax, bx):
        return self.compute_backward_posterior(az, bz, ax, bx)[0]      # :81-103

    def compute_forward_mean(self, az, bz, ax, bx):
        return self.compute_forward_posterior(az, bz, ax, bx)[0]       # :105-109

    def compute_backward_posterior(self, az, bz, ax, bx):
        t, a, _, _, r, v = self._run(_lib.BWD, az, bz, ax, bx, want_posterior=True)
        return self._ret(r, t), self._ret_scal(v, a, t)

    def compute_forward_posterior(self, az, bz, ax, bx):
        t, a, _, _, r, v = self._run(_lib.FWD, az, bz, ax, bx, want_posterior=True)
        return self._ret(r, t), self._ret_scal(v, a, t)

    def compute_forward_message(self, az, bz, ax, bx, **kw):
        t, a, a_new, b_new, _, _ = self._run(_lib.FWD, az, bz, ax, bx, **kw)
        return self._ret_scal(a_new, a, t), self._ret(b_new, t)

    def compute_backward_message(self, az, bz, ax, bx, **kw):
        t, a, a_new, b_new, _, _ = self._run(_lib.BWD, az, bz, ax, bx, **kw)
        return self._ret_scal(a_new, a, t), self._ret(b_new, t)

    @staticmethod
    def _ret_scal(dev, like, vec_like):
        if like.was_host and vec_like.was_host:
            h = dev.get()
            return float(h[0]) if (like.scalar_like and getattr(vec_like, "_one", False)) else h
        return dev


def _fd1(N):
    "forward first-difference stencil (tramp/utils/conv_filters.py:29-34)"
    f = np.zeros(N)
    f[0], f[1] = -1, 1
    return f


def _fd2(N):
    "second-difference stencil (conv_filters.py:37-42)"
    f = np.zeros(N)
    f[0] = -2
    f[1] = f[-1] = 1
    return f


def _along_axis(stencil, axis, shape):
    "place a 1-D stencil on `axis` at the origin of the other axes (conv_filters.py:55-78)"
    f = np.zeros(shape)
    idx = [0] * len(shape)
    idx[axis] = slice(None)
    f[tuple(idx)] = stencil(shape[axis])
    return f


def gaussian_filter(sigma, N):
    "periodic Gaussian blur with unit DC gain (conv_filters.py:45-52)"
    freq = np.fft.fftfreq(N)
    return np.real(np.fft.ifft(np.exp(-2 * (np.pi * sigma)**2 * freq**2)))


def differential_filter(shape, D1, D2=None):
    "D = D1 . d/dx + D2 . d2/dx2 (conv_filters.py:81-90)"
    d = len(shape)
    D2 = D2 or np.zeros(d)
    return sum(D1[ax] * _along_axis(_fd1, ax, shape) for ax in range(d)) + \
        sum(D2[ax] * _along_axis(_fd2, ax, shape) for ax in range(d))


def laplacian_filter(shape):
    return sum(_along_axis(_fd2, ax, shape) for ax in range(len(shape)))     # conv_filters.py:93-99


class DifferentialChannel(ConvChannel):
    "conv_channel.py:166-176"

    def __init__(self, D1, D2, shape, real=True, **kw):
        super().__init__(filter=differential_filter(shape=tuple(shape), D1=D1, D2=D2), real=real, **kw)
        self.D1, self.D2 = D1, D2


class LaplacianChannel(ConvChannel):
    "conv_channel.py:179-186"

    def __init__(self, shape, real=True, **kw):
        super().__init__(filter=laplacian_filter(tuple(shape)), real=real, **kw)


class Blur1DChannel(ConvChannel):
    "conv_channel.py:189-194"

    def __init__(self, sigma, N, real=True, **kw):
        super().__init__(filter=gaussian_filter(sigma=sigma, N=N), real=real, **kw)
        self.sigma = sigma


class Blur2DChannel(ConvChannel):
    "conv_channel.py:197-208"

    def __init__(self, sigma, shape, real=True, **kw):
        if len(sigma) != 2:
            raise ValueError("sigma must be a length 2 array")
        if len(shape) != 2:
            raise ValueError("shape must be a length 2 tuple")
        f = np.outer(gaussian_filter(sigma=sigma[0], N=shape[0]), gaussian_filter(sigma=sigma[1], N=shape[1]))
        super().__init__(filter=f, real=real, **kw)
        self.sigma = sigma


def gradient_filters(shape):
    "one forward-difference filter per axis, (d,) + shape (conv_filters.py:102-119)"
    return np.stack([_along_axis(_fd1, ax, tuple(shape)) for ax in range(len(shape))])


class GradientChannel(Channel):
    """Gradient channel x = grad z (circular forward differences along each of the d axes).

    tramp/channels/linear/gradient_channel.py:10-118 (``real=True``): z has ``shape``, x has
    ``(d,) + shape``; diagonal in the Fourier basis on the summed spectrum of the d filters
    (csrc/conv.hip, rocFFT transforms + this library's spectral and update kernels).  Device
    vectors: z [batch, n], x [batch, d n].
    """

    def __init__(self, shape, real=True, dtype=np.float64, ctx=None):
        shape = tuple(int(s) for s in shape)
        if not 1 <= len(shape) <= 3:
            raise ValueError("shape must have 1 to 3 axes")
        self.d = len(shape)
        self.shape = shape
        self.real = real
        self.repr_init()
        self._init_device(dtype, ctx)
        self.filter = gradient_filters(shape)
        self.N = int(np.prod(shape))
        self.Nz, self.Nx = self.N, self.d * self.N
        ctx = self.ctx
        fd = ctx.array(self.filter.reshape(-1), np.float64)
        dims = (C.c_size_t * self.d)(*shape)
        h = C.c_void_p()
        ctx._call("tramp_grad_create", self._dt, self.d, dims, fd.ptr, C.byref(h))
        self._handle = h
        cnt, mean = C.c_size_t(), C.c_double()
        ctx._call("tramp_grad_spectrum", h, None, C.byref(cnt), C.byref(mean))
        self._n_half = cnt.value
        self.spectrum_mean = mean.value

    def __del__(self):
        try:
            self.ctx.release("tramp_grad_destroy", self._handle)
        except Exception:
            pass
        self._handle = None

    def _vec(self, b, x_side):
        if isinstance(b, DeviceArray):
            return VecArg(self.ctx, b, self.dtype)
        full = ((self.d,) + self.shape) if x_side else self.shape
        n = self.Nx if x_side else self.Nz
        h = np.asarray(b, dtype=self.dtype)
        nd = len(full)
        if h.shape[-nd:] != full or h.ndim not in (nd, nd + 1):
            raise ValueError(f"expected an array of shape {full} or (batch,)+{full}, got {h.shape}")
        arg = VecArg(self.ctx, h.reshape(-1, n) if h.ndim == nd + 1 else h.reshape(n), self.dtype)
        arg.host_shape = h.shape
        arg._one = h.ndim == nd
        return arg

    def _ret(self, dev, like):
        """real=False only drops the reference's np.real() (gradient_channel.py:32-33, 73-74, 82-83): with the real
        inputs the channel takes, host results come back as complex arrays whose imaginary part is zero (round-off in
        the reference, exactly zero here)."""
        out = ConvChannel._ret(dev, like)
        return out.astype(np.complex128) if (not self.real and like.was_host) else out

    _ret_scal = staticmethod(ConvChannel._ret_scal)

    def convolve(self, z):
        "gradient_channel.py:26-34"
        zz = self._vec(z, False)
        x = self.ctx.empty((zz.batch, self.Nx), self.dtype)
        self.ctx._call("tramp_grad_apply", self._handle, zz.batch, zz.dev.ptr, x.ptr)
        self._t_key = self._u_key = None                               # the call reuses the transform buffers
        if zz.was_host:
            out = x.get().reshape(((zz.batch,) if not zz._one else ()) + (self.d,) + self.shape)
            return out if self.real else out.astype(np.complex128)
        return x

    def sample(self, Z):
        return self.convolve(Z)

    def second_moment(self, tau_z):
        return tau_z * self.spectrum_mean / self.d                     # :42-43

    def device_variance(self, direction, az, ax):
        ctx = self.ctx
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        a_z = ctx.scalars(az_h)
        a_x = ctx.scalars(ax, az_h.shape[0])
        outs = [ctx.empty((az_h.shape[0],), np.float64) for _ in range(3)]
        ctx._call("tramp_grad_variance", self._handle, direction, az_h.shape[0], a_z.ptr, a_x.ptr,
                  *[o.ptr for o in outs])
        return [o.get() for o in outs]

    def compute_n_eff(self, az, ax):
        return float(self.device_variance(_lib.BWD, az, ax)[2][0])     # :45-55

    def compute_backward_variance(self, az, ax):
        assert az > 0                                                  # :85-89
        return float(self.device_variance(_lib.BWD, az, ax)[0][0])

    def compute_forward_variance(self, az, ax):
        return float(self.device_variance(_lib.FWD, az, ax)[0][0])     # :91-97

    def compute_backward_error(self, az, ax, tau_z):
        return self.compute_backward_variance(az, ax)                  # :111-113

    def compute_forward_error(self, az, ax, tau_z):
        return self.compute_forward_variance(az, ax)                   # :115-117

    def _run(self, direction, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
             old_b=None, want_posterior=False):
        ctx = self.ctx
        z, x = self._vec(bz, False), self._vec(bx, True)
        if z.n != self.Nz or x.n != self.Nx or z.batch != x.batch:
            raise ValueError(f"expected bz [batch,{self.Nz}] and bx [batch,{self.Nx}], got {z.dev.shape} and {x.dev.shape}")
        a_z = ScalArg(ctx, az, z.batch)
        a_x = ScalArg(ctx, ax, z.batch)
        tgt, a_t = (x, a_x) if direction == _lib.FWD else (z, a_z)
        a_new = out_a if out_a is not None else ctx.empty((z.batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((z.batch, tgt.n), self.dtype)
        r = ctx.empty((z.batch, tgt.n), self.dtype) if want_posterior else None
        v = ctx.empty((z.batch,), np.float64) if want_posterior else None
        # the transforms of device-resident messages that have not been rewritten since the last call stay in the handle
        reuse = 0
        kz, kx = LinearChannel._proj_key(bz, z), LinearChannel._proj_key(bx, x)
        if kz is not None and kz == getattr(self, "_t_key", None):
            reuse |= _lib.REUSE_T
        if kx is not None and kx == getattr(self, "_u_key", None):
            reuse |= _lib.REUSE_U
        self._t_key, self._u_key = kz, kx
        ctx._call("tramp_grad_message", self._handle, direction, z.batch, a_z.dev.ptr, z.dev.ptr,
                  a_x.dev.ptr, x.dev.ptr, float(damping or 0.0),
                  old_a.ptr if old_a is not None else None, old_b.ptr if old_b is not None else None,
                  a_new.ptr, b_new.ptr, r.ptr if r is not None else None, v.ptr if v is not None else None, reuse)
        return tgt, a_t, a_new, b_new, r, v

    compute_backward_mean = ConvChannel.compute_backward_mean
    compute_forward_mean = ConvChannel.compute_forward_mean
    compute_backward_posterior = ConvChannel.compute_backward_posterior
    compute_forward_posterior = ConvChannel.compute_forward_posterior
    compute_forward_message = ConvChannel.compute_forward_message
    compute_backward_message = ConvChannel.compute_backward_message


class _UnitaryChannel(Channel):
    """Shared message plumbing of the unitary channels (DFT, rotation): the posterior is not
    needed - (a, b) travel through the change of basis (a unchanged, b transformed)."""

    def second_moment(self, tau_z):
        return tau_z

    def compute_forward_state_evolution(self, az, ax, tau_z):
        return az

    def compute_backward_state_evolution(self, az, ax, tau_z):
        return ax

    def compute_forward_error(self, az, ax, tau_z):
        return 1.0 / (az + ax)

    def compute_backward_error(self, az, ax, tau_z):
        return 1.0 / (az + ax)

    def _finish(self, a_in, src, a_new, b_new, out_a, out_b):
        if out_a is not None:
            return out_a, out_b
        if src.was_host:
            a = a_new.get()
            a = float(a[0]) if (a_in.was_host and a_in.scalar_like and len(src.host_shape) == 1) else a
            return a, b_new.get().reshape(self._host_out_shape(src))
        return a_new, b_new


class DFTChannel(_UnitaryChannel):
    """x = fftn(z, norm="ortho") (tramp/channels/linear/dft_channel.py:7-77), z real (``real=True``) or complex.

    A complex vector is the planar real array [2, n] as in the reference; on the device
    [batch, 2 n].  ``shape`` is only needed for transforms over more than one axis."""

    def __init__(self, real=True, shape=None, dtype=np.float64, ctx=None):
        self.real = bool(real)
        self.shape = None if shape is None else tuple(int(v) for v in np.atleast_1d(shape))
        self.repr_init()
        self._init_device(dtype, ctx)
        self._handle = None
        self._n = None

    def _ensure(self, n):
        if self._handle is not None and self._n == n:
            return
        dims = self.shape if self.shape is not None else (n,)
        if int(np.prod(dims)) != n:
            raise ValueError(f"message of {n} components does not match shape {dims}")
        self.__del__()
        arr = (C.c_size_t * len(dims))(*dims)
        h = C.c_void_p()
        self.ctx._call("tramp_dft_create", self._dt, len(dims), arr, C.byref(h))
        self._handle, self._n = h, n
        if not self.real:
            self.ctx._call("tramp_dft_set_complex_input", h, 1)

    def __del__(self):
        try:
            self.ctx.release("tramp_dft_destroy", self._handle)
        except Exception:
            pass
        self._handle = None

    def _host_out_shape(self, src):
        return self._out_shape

    def _vec(self, b, planar):
        if isinstance(b, DeviceArray):
            return VecArg(self.ctx, b, self.dtype)
        h = np.asarray(b, dtype=self.dtype)
        core = h.shape[-(len(self.shape) if self.shape else 1):]
        nd = len(core) + (1 if planar else 0)
        if h.ndim not in (nd, nd + 1):
            raise ValueError(f"unexpected message shape {h.shape}")
        n_tot = int(np.prod(h.shape[-nd:]))
        arg = VecArg(self.ctx, h.reshape(-1, n_tot) if h.ndim == nd + 1 else h.reshape(n_tot), self.dtype)
        lead = h.shape[:-nd]
        self._lead, self._core = lead, core
        return arg

    def _message(self, direction, a_in, b_in, out_a, out_b, damping, old_a, old_b):
        ctx = self.ctx
        planar_in = direction == _lib.BWD or not self.real          # x is always complex, z when real=False
        planar_out = direction == _lib.FWD or not self.real
        src = self._vec(b_in, planar=planar_in)
        if not isinstance(b_in, DeviceArray):
            self._out_shape = self._lead + ((2,) + self._core if planar_out else self._core)
        n = src.n // 2 if planar_in else src.n
        self._ensure(n)
        a = ScalArg(ctx, a_in, src.batch)
        a_new = out_a if out_a is not None else ctx.empty((src.batch,), np.float64)
        n_out = 2 * n if planar_out else n
        b_new = out_b if out_b is not None else ctx.empty((src.batch, n_out), self.dtype)
        ctx._call("tramp_dft_message", self._handle, direction, src.batch, a.dev.ptr, src.dev.ptr,
                  float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr)
        return self._finish(a, src, a_new, b_new, out_a, out_b)

    def sample(self, Z):
        return self.compute_forward_message(1.0, Z, 0.0, None)[1]     # dft_channel.py:32-38

    def compute_forward_message(self, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
                                old_b=None):
        return self._message(_lib.FWD, az, bz, out_a, out_b, damping, old_a, old_b)      # :46-53

    def compute_backward_message(self, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
                                 old_b=None):
        return self._message(_lib.BWD, ax, bx, out_a, out_b, damping, old_a, old_b)      # :55-64


class RotationChannel(_UnitaryChannel):
    """x = R z with R orthogonal (tramp/channels/linear/rotation_channel.py:5-60): one pass of
    the dense matrix kernels per message, b_new = R bz forward, R^T bx backward."""

    def __init__(self, R, name="R", dtype=np.float64, ctx=None):
        R = np.asarray(R)
        if R.ndim != 2 or R.shape[0] != R.shape[1]:
            raise ValueError(f"R.shape = {R.shape}")
        if not np.allclose(R @ R.T, np.identity(R.shape[0])):
            raise ValueError("R not a rotation")
        self.name = name
        self.N = R.shape[0]
        self.repr_init()
        self._init_device(dtype, ctx)
        self.R = R
        self._R_dev = self.ctx.array(np.ascontiguousarray(R, dtype=self.dtype), self.dtype)

    def _host_out_shape(self, src):
        return src.host_shape

    def sample(self, Z):
        return self.R @ Z

    def _message(self, direction, a_in, b_in, out_a, out_b, damping, old_a, old_b):
        ctx = self.ctx
        src = VecArg(ctx, b_in, self.dtype)
        if src.n != self.N:
            raise ValueError(f"expected messages of {self.N} components, got {src.n}")
        a = ScalArg(ctx, a_in, src.batch)
        d = float(damping or 0.0)
        mv = "tramp_matvec_nt" if direction == _lib.FWD else "tramp_matvec_nn"
        if out_a is not None and (not d or (old_a.ptr == out_a.ptr and old_b.ptr == out_b.ptr)):
            # scheduler path: no per-call temporaries (a captured graph replays these launches); the rotated
            # message lands in a scratch kept by the channel, then out <- (1 - d) new + d out in place
            if not d:
                ctx._call(mv, self._dt, 0, self.N, self.N, src.batch, self._R_dev.ptr, src.dev.ptr, out_b.ptr)
                out_a.copy_from(a.dev)
            else:
                scratch = getattr(self, "_scratch", None)
                if scratch is None or scratch.shape != (src.batch, self.N):
                    scratch = self._scratch = ctx.empty((src.batch, self.N), self.dtype)
                ctx._call(mv, self._dt, 0, self.N, self.N, src.batch, self._R_dev.ptr, src.dev.ptr, scratch.ptr)
                ctx._call("tramp_damp", self._dt, self.N, src.batch, 1.0 - d, a.dev.ptr, scratch.ptr, out_a.ptr,
                          out_b.ptr)
            return out_a, out_b
        b_tmp = ctx.empty((src.batch, self.N), self.dtype)
        ctx._call(mv, self._dt, 0, self.N, self.N, src.batch, self._R_dev.ptr, src.dev.ptr, b_tmp.ptr)
        a_tmp = ctx.empty((src.batch,), np.float64)
        a_tmp.copy_from(a.dev)
        if d:
            ctx._call("tramp_damp", self._dt, self.N, src.batch, d, old_a.ptr, old_b.ptr, a_tmp.ptr, b_tmp.ptr)
        if out_a is not None:
            out_a.copy_from(a_tmp)
            out_b.copy_from(b_tmp)
        return self._finish(a, src, a_tmp, b_tmp, out_a, out_b)

    def compute_forward_message(self, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
                                old_b=None):
        return self._message(_lib.FWD, az, bz, out_a, out_b, damping, old_a, old_b)      # :32-36

    def compute_backward_message(self, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
                                 old_b=None):
        return self._message(_lib.BWD, ax, bx, out_a, out_b, damping, old_a, old_b)      # :38-42


class ReshapeChannel(_UnitaryChannel):
    """x = z.reshape(next_shape) (tramp/channels/shape/reshape_channel.py:4-37): the messages pass through
    unchanged.  Device vectors are flat [batch, n] on both sides, so this is a copy (and the hook on which
    constant damping of the edge is applied)."""

    def __init__(self, prev_shape, next_shape, dtype=np.float64, ctx=None):
        as_tuple = lambda s: (int(s),) if np.isscalar(s) else tuple(int(v) for v in s)      # noqa: E731
        self.prev_shape, self.next_shape = as_tuple(prev_shape), as_tuple(next_shape)
        if int(np.prod(self.prev_shape)) != int(np.prod(self.next_shape)):
            raise ValueError(f"cannot reshape {self.prev_shape} into {self.next_shape}")
        self.N = int(np.prod(self.prev_shape))
        self.repr_init()
        self._init_device(dtype, ctx)

    def sample(self, Z):
        return np.asarray(Z).reshape(self.next_shape)

    def _vec(self, b, shape):
        if isinstance(b, DeviceArray):
            return VecArg(self.ctx, b, self.dtype)
        h = np.asarray(b, dtype=self.dtype)
        nd = len(shape)
        if h.shape[-nd:] != shape or h.ndim not in (nd, nd + 1):
            raise ValueError(f"expected an array of shape {shape} or (batch,)+{shape}, got {h.shape}")
        arg = VecArg(self.ctx, h.reshape(-1, self.N) if h.ndim == nd + 1 else h.reshape(self.N), self.dtype)
        arg.host_shape = h.shape
        arg._lead = h.shape[:-nd]
        return arg

    def _message(self, direction, a_in, b_in, out_a, out_b, damping, old_a, old_b):
        ctx = self.ctx
        src_shape, dst_shape = (self.prev_shape, self.next_shape) if direction == _lib.FWD else \
            (self.next_shape, self.prev_shape)
        src = self._vec(b_in, src_shape)
        a = ScalArg(ctx, a_in, src.batch)
        d = float(damping or 0.0)
        if out_a is not None and (not d or (old_a.ptr == out_a.ptr and old_b.ptr == out_b.ptr)):
            # scheduler path, no temporaries (safe under graph capture): in-place damping with the roles
            # swapped, out <- (1 - d) src + d out
            if d:
                ctx._call("tramp_damp", self._dt, self.N, src.batch, 1.0 - d, a.dev.ptr, src.dev.ptr, out_a.ptr,
                          out_b.ptr)
            else:
                out_a.copy_from(a.dev)
                out_b.copy_from(src.dev)
            return out_a, out_b
        b_tmp = ctx.empty((src.batch, self.N), self.dtype)
        b_tmp.copy_from(src.dev)
        a_tmp = ctx.empty((src.batch,), np.float64)
        a_tmp.copy_from(a.dev)
        if d:
            ctx._call("tramp_damp", self._dt, self.N, src.batch, d, old_a.ptr, old_b.ptr, a_tmp.ptr, b_tmp.ptr)
        if out_a is not None:
            out_a.copy_from(a_tmp)
            out_b.copy_from(b_tmp)
            return out_a, out_b
        if src.was_host:
            av = a_tmp.get()
            av = float(av[0]) if (a.was_host and a.scalar_like and not src._lead) else av
            return av, b_tmp.get().reshape(src._lead + dst_shape)
        return a_tmp, b_tmp

    def compute_forward_message(self, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
                                old_b=None):
        return self._message(_lib.FWD, az, bz, out_a, out_b, damping, old_a, old_b)      # :27-28

    def compute_backward_message(self, az, bz, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None,
                                 old_b=None):
        return self._message(_lib.BWD, ax, bx, out_a, out_b, damping, old_a, old_b)      # :30-31


class UnitaryChannel(RotationChannel):
    """x = U z with U unitary (tramp/channels/linear/unitary_channel.py:16-97).  A complex vector is
    the planar real array (2, N) = (real parts, imaginary parts) as in the reference, so U acts as the
    2N x 2N rotation [[Re U, -Im U], [Im U, Re U]] on the flattened array: the same dense matrix
    kernels as RotationChannel (device vectors [batch, 2 N])."""

    def __init__(self, U, name="U", dtype=np.float64, ctx=None):
        U = np.asarray(U)
        if U.ndim != 2 or U.shape[0] != U.shape[1]:
            raise ValueError(f"U.shape = {U.shape}")
        if not np.allclose(U @ U.conj().T, np.identity(U.shape[0])):
            raise ValueError("U not unitary")
        self.U = U
        self.Nc = U.shape[0]
        R = np.block([[U.real, -U.imag], [U.imag, U.real]])
        super().__init__(R, name=name, dtype=dtype, ctx=ctx)

    def sample(self, Z):
        Z = np.asarray(Z)
        X = self.U @ (Z[0] + 1j * Z[1])
        return np.stack([X.real, X.imag])

    def _message(self, direction, a_in, b_in, out_a, out_b, damping, old_a, old_b):
        if isinstance(b_in, DeviceArray):
            return super()._message(direction, a_in, b_in, out_a, out_b, damping, old_a, old_b)
        h = np.asarray(b_in)
        if h.shape[-2:] != (2, self.Nc) or h.ndim not in (2, 3):
            raise ValueError(f"expected a planar complex array (2, {self.Nc}) or (batch, 2, {self.Nc}), got {h.shape}")
        flat = h.reshape(2 * self.Nc) if h.ndim == 2 else h.reshape(-1, 2 * self.Nc)
        a_new, b_new = super()._message(direction, a_in, flat, out_a, out_b, damping, old_a, old_b)
        return a_new, (b_new.reshape(h.shape) if isinstance(b_new, np.ndarray) else b_new)


class AnalyticalLinearChannel(Channel):
    "SE-only linear channel with closed-form spectrum transforms (analytical_linear_channel.py:8-62)"

    def __init__(self, ensemble, name="W"):
        self.name = name
        self.alpha = ensemble.alpha
        self.repr_init()
        self.ensemble = ensemble

    def second_moment(self, tau_z):
        return tau_z * (self.ensemble.mean_spectrum / self.alpha)

    def _k15(self, direction, az, ax):
        "kernel K15 (tramp_se_mp_channel) on arrays of points: (v, n_eff)"
        if not hasattr(self.ensemble, "compute_F"):
            raise NotImplementedError("only the Marchenko-Pastur closed form is on the HIP path")
        ctx = default_context()
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        n = az_h.shape[0]
        ax_h = np.ascontiguousarray(np.broadcast_to(np.asarray(ax, dtype=np.float64), (n,)))
        al, ms = np.full(n, float(self.alpha)), np.full(n, float(self.ensemble.mean_spectrum))
        v, ne = np.empty(n), np.empty(n)
        ctx._call("tramp_se_mp_channel", direction, n, al.ctypes.data, ms.ctypes.data, az_h.ctypes.data,
                  ax_h.ctypes.data, v.ctypes.data, ne.ctypes.data)
        return (float(v[0]), float(ne[0])) if np.ndim(az) == 0 else (v, ne)

    def compute_n_eff(self, az, ax):
        return self._k15(_lib.FWD, az, ax)[1]                   # analytical_linear_channel.py:28-37

    def compute_backward_error(self, az, ax, tau_z):
        return self._k15(_lib.BWD, az, ax)[0]                   # :42-46

    def compute_forward_error(self, az, ax, tau_z):
        return self._k15(_lib.FWD, az, ax)[0]                   # :48-53


class MarchenkoPasturChannel(AnalyticalLinearChannel):
    "analytical_linear_channel.py:65-90"

    def __init__(self, alpha, name="W"):
        super().__init__(ensemble=MarchenkoPasturEnsemble(alpha=alpha), name=name)


class ModulusChannel(Channel):
    """Modulus channel x = |z| (tramp/channels/activation/modulus_channel.py:9-43): a GENERATIVE factor only, as in the
    reference (its posteriors raise NotImplementedError there too); ``Model.to_observed`` turns it into ModulusLikelihood
    (models/dag_algebra.py:38-39)."""

    def __init__(self, dtype=np.float64, ctx=None):
        self.repr_init()
        self._init_device(dtype, ctx)

    def sample(self, Z):
        "We assume Z[0] = Z.real and Z[1] = Z.imag"
        Z = np.asarray(Z)
        return np.absolute(Z[0] + 1j * Z[1])

    def second_moment(self, tau_z):
        return 2 * tau_z

    def compute_forward_posterior(self, az, bz, ax, bx):
        raise NotImplementedError

    def compute_backward_posterior(self, az, bz, ax, bx):
        raise NotImplementedError


CHANNEL_CLASSES = {
    "gaussian": GaussianChannel,
    "duplicate": DuplicateChannel,
    "linear": LinearChannel,
    "complex_linear": ComplexLinearChannel,
    "modulus": ModulusChannel,
    "analytical": AnalyticalLinearChannel,
    "marchenko": MarchenkoPasturChannel,
    "sgn": SgnChannel,
    "abs": AbsChannel,
    "relu": ReluChannel,
    "l-relu": LeakyReluChannel,
    "a-abs": AsymmetricAbsChannel,
    "h-tanh": HardTanhChannel,
    "h-sigm": HardSigmoidChannel,
    "door": SymmetricDoorChannel,
    "conv": ConvChannel,
    "blur_1d": Blur1DChannel,
    "blur_2d": Blur2DChannel,
    "diff": DifferentialChannel,
    "laplacian": LaplacianChannel,
    "dft": DFTChannel,
    "rotation": RotationChannel,
    "gradient": GradientChannel,
    "unitary": UnitaryChannel,
    "reshape": ReshapeChannel,
    "bias": BiasChannel,
    "sum": SumChannel,
}


def get_channel(channel_type, **kwargs):
    "tramp/channels/__init__.py:68-70"
    return CHANNEL_CLASSES[channel_type](**kwargs)
