"""Node protocol of the factor graph (host side).

Mirror of tramp/base.py: ``Variable`` / ``Factor`` with the message parse/emit
methods the scheduler calls (``forward_message`` / ``backward_message`` /
``forward_state_evolution`` / ``backward_state_evolution``, tramp/base.py:329-413)
and the scalar message update (``compute_a_new`` :245-248).  The N-dimensional
part of every update runs in HIP kernels behind the C ABI; this module only moves
handles around.

Messages are ``(source, target, data)`` triples like the reference's, with
``data = dict(a=..., b=..., direction="fwd"|"bwd")``.  ``a`` is a ``DeviceArray``
of ``batch`` doubles and ``b`` a ``DeviceArray`` [batch, n] in resident mode, or
Python/NumPy values in strict (NumPy in, NumPy out) mode.
"""
import numpy as np

from . import _lib
from .device import DeviceArray, default_context, dtype_code

AMIN = 1e-11   # tramp/base.py:238-239
AMAX = 1e+11


def inv(v):
    "tramp/base.py:44-46"
    return 1 / np.maximum(v, 1e-20)


def filter_message(message, direction):
    return [(s, t, d) for s, t, d in message if d["direction"] == direction]


class ReprMixin:
    def repr_init(self):
        self._repr_kwargs = {k: v for k, v in self.__dict__.items() if not k.startswith("_")}

    def __repr__(self):
        kw = getattr(self, "_repr_kwargs", {})
        args = ",".join(f"{k}={v if np.isscalar(v) or v is None else type(v).__name__}"
                        for k, v in kw.items())
        return f"{self.__class__.__name__}({args})"


class VecArg:
    """Normalised view of a ``b``-type argument: device buffer [batch, n]."""

    def __init__(self, ctx, b, dtype):
        self.was_host = not isinstance(b, DeviceArray)
        if self.was_host:
            h = np.asarray(b, dtype=dtype)
            if h.ndim == 0 or h.ndim > 2:
                raise ValueError(f"b must be a vector [n] or an ensemble [batch, n], got shape {h.shape}")
            self.host_shape = h.shape
            self.dev = ctx.array(np.atleast_2d(h), dtype)
        else:
            if b.dtype != np.dtype(dtype):
                raise TypeError(f"device vector has dtype {b.dtype}, factor computes in {np.dtype(dtype)}")
            self.host_shape = b.shape
            self.dev = b if b.ndim == 2 else b.reshape(1, b.shape[0])
        self.batch, self.n = self.dev.shape


class ScalArg:
    """Normalised view of an ``a``-type argument: double[batch] on the device."""

    def __init__(self, ctx, a, batch):
        self.was_host = not isinstance(a, DeviceArray)
        if self.was_host:
            self.scalar_like = np.ndim(a) == 0
            self.dev = ctx.scalars(a, batch)
        else:
            self.scalar_like = False
            if a.size != batch or a.dtype != np.float64:
                raise ValueError(f"a must hold {batch} doubles, got {a.shape} {a.dtype}")
            self.dev = a


def out_vec(ctx, like: VecArg, dtype):
    return ctx.empty((like.batch, like.n), dtype)


def ret_vec(dev, like: VecArg):
    "Device result -> what the caller handed in (NumPy array or DeviceArray)."
    if like.was_host:
        return dev.get().reshape(like.host_shape)
    return dev if len(like.host_shape) == 2 else dev.reshape(like.host_shape)


def ret_scal(dev, like: ScalArg, vec_like: VecArg):
    if like.was_host and vec_like.was_host:
        h = dev.get()
        if like.scalar_like and len(vec_like.host_shape) == 1:
            return float(h[0])
        return h
    return dev


class Variable(ReprMixin):
    def __init__(self, id, n_prev, n_next):
        self.id = id
        self.n_prev = n_prev
        self.n_next = n_next
        self.repr_init()

    def __matmul__(self, other):
        from .models import DAG
        return DAG(self) @ other

    def __add__(self, other):
        from .models import DAG
        return DAG(self) + other


class Factor(ReprMixin):
    """Base of priors / channels / likelihoods (tramp/base.py:236-464)."""
    AMAX = AMAX
    AMIN = AMIN
    dtype = np.float64
    id = None

    def _init_device(self, dtype=np.float64, ctx=None):
        self.dtype = np.dtype(dtype)
        self._dt = dtype_code(dtype)
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def compute_a_new(self, v, a):
        "tramp/base.py:245-248"
        return np.clip(inv(v) - a, self.AMIN, self.AMAX)

    # ``isotropic=False``: a is an array like b, variances stay per component, compute_ab_new clips element-wise
    # (tramp/base.py:250-255) - kernel K1d..K7d, tramp_diag_factor_p.  Subclasses set _diag_which and _diag_params().
    _diag_which = None

    def _diag_params(self):
        return ()

    def _diag(self, a, b, y=None):
        "(a_new, b_new, r, v) as host arrays of b's shape"
        import ctypes as C
        if self._diag_which is None:
            raise NotImplementedError(f"diagonal beliefs are not built for {type(self).__name__}")
        ctx = self.ctx
        bh = np.asarray(b, dtype=self.dtype)
        bd = ctx.array(bh)
        ad = ctx.array(np.broadcast_to(np.asarray(a, dtype=self.dtype), bh.shape).copy())
        yd = ctx.array(np.broadcast_to(np.asarray(y, dtype=self.dtype), bh.shape).copy()) if y is not None else None
        par = [float(x) for x in self._diag_params()]
        arr = (C.c_double * max(len(par), 1))(*par)
        outs = [ctx.empty(bh.shape, self.dtype) for _ in range(4)]
        ctx._call("tramp_diag_factor_p", self._dt, int(self._diag_which), bd.size, ad, bd, yd, arr, len(par), *outs)
        return [o.get() for o in outs]

    def _logz(self, kind, a0, v0, a1=None, v1=None, params=(0.0, 0.0, 0.0), mean=True):
        """Log-partition through tramp_log_partition: the kernel returns the sum
        over the components; ``mean`` divides by n as the reference's priors / likelihoods do."""
        ctx = self.ctx
        if not getattr(self, "isotropic", True) and np.ndim(a0) > 0 and not isinstance(v0, DeviceArray):
            # diagonal beliefs: every component has its own precision - n instances of one component each,
            # then the reference's A.mean() (e.g. gauss_bernoulli_prior.py:79-83) on the host
            vh = np.asarray(v0, dtype=self.dtype).reshape(-1, 1)
            ah = np.broadcast_to(np.asarray(a0, dtype=np.float64).ravel(), (vh.shape[0],)).copy()
            wh = None
            if v1 is not None:
                wh = (v1.get() if isinstance(v1, DeviceArray) else np.asarray(v1, dtype=self.dtype)).reshape(-1, 1)
            iso, self.isotropic = self.isotropic, True
            try:
                per = np.atleast_1d(self._logz(kind, ah, vh, a1, wh, params, mean=False))
            finally:
                self.isotropic = iso
            return float(per.mean()) if mean else float(per.sum())
        v = VecArg(ctx, v0, self.dtype)
        a = ScalArg(ctx, a0, v.batch)
        w = VecArg(ctx, v1, self.dtype) if v1 is not None else None
        aa = ScalArg(ctx, a1, v.batch) if a1 is not None else None
        if w is not None and (w.batch, w.n) != (v.batch, v.n):
            raise ValueError(f"shapes {v.dev.shape} and {w.dev.shape} differ")
        out = ctx.empty((v.batch,), np.float64)
        p = [float(x) for x in params] + [0.0] * (3 - len(params))
        ctx._call("tramp_log_partition", self._dt, kind, v.n, v.batch, p[0], p[1], p[2], a.dev.ptr, v.dev.ptr,
                  aa.dev.ptr if aa is not None else None, w.dev.ptr if w is not None else None, out.ptr)
        res = out.get()
        if mean:
            res = res / v.n
        if a.was_host and a.scalar_like and v.was_host and len(v.host_shape) == 1:
            return float(res[0])
        return res

    def _logz_regions(self, mode, regions, az, bz, v1, ax=None, mean=False):
        """Log-partition of a factor given as a table of linear regions [(zmin, zmax, slope, x0)] in the reference's
        order, through tramp_region_log_partition: mode 0 = channel (v1 = bx, summed,
        piecewise_linear_channel.py:55-58), mode 1 = likelihood (v1 = y, mean, piecewise_linear_likelihood.py:216-219)."""
        ctx = self.ctx
        # the table as the kernels of the working precision hold it (a kink at float32(1e-4), not at 1e-4)
        reg = np.ascontiguousarray(np.asarray(regions, dtype=self.dtype).astype(np.float64).reshape(-1, 4))
        if not getattr(self, "isotropic", True) and np.ndim(az) > 0 and not isinstance(bz, DeviceArray):
            # diagonal beliefs: n instances of one component each, then the reference's mean / sum on the host
            vh = np.asarray(bz, dtype=self.dtype).reshape(-1, 1)
            ah = np.broadcast_to(np.asarray(az, dtype=np.float64).ravel(), (vh.shape[0],)).copy()
            wh = (v1.get() if isinstance(v1, DeviceArray) else np.asarray(v1, dtype=self.dtype)).reshape(-1, 1)
            axh = None if ax is None else np.broadcast_to(np.asarray(ax, dtype=np.float64).ravel(), (vh.shape[0],)).copy()
            iso, self.isotropic = self.isotropic, True
            try:
                per = np.atleast_1d(self._logz_regions(mode, reg, ah, vh, wh, ax=axh, mean=False))
            finally:
                self.isotropic = iso
            return float(per.mean()) if mean else float(per.sum())
        v = VecArg(ctx, bz, self.dtype)
        a = ScalArg(ctx, az, v.batch)
        w = VecArg(ctx, v1, self.dtype)
        aa = ScalArg(ctx, ax, v.batch) if ax is not None else None
        if (w.batch, w.n) != (v.batch, v.n):
            raise ValueError(f"shapes {v.dev.shape} and {w.dev.shape} differ")
        out = ctx.empty((v.batch,), np.float64)
        ctx._call("tramp_region_log_partition", self._dt, int(mode), reg.shape[0], reg.ctypes.data, v.n, v.batch,
                  a.dev.ptr, v.dev.ptr, aa.dev.ptr if aa is not None else None, w.dev.ptr, out.ptr)
        res = out.get()
        if mean:
            res = res / v.n
        if a.was_host and a.scalar_like and v.was_host and len(v.host_shape) == 1:
            return float(res[0])
        return res

    def __matmul__(self, other):
        from .models import DAG
        return DAG(self) @ other

    def __add__(self, other):
        from .models import DAG
        return DAG(self) + other

    # -- message parse / emit (tramp/base.py:275-365)
    def _parse_message_ab(self, message):
        z = filter_message(message, "fwd")
        x = filter_message(message, "bwd")
        assert len(z) == self.n_prev and len(x) == self.n_next
        az = bz = z_source = ax = bx = x_source = None
        if self.n_prev == 1:
            z_source, _, d = z[0]
            az, bz = d["a"], d["b"]
        if self.n_next == 1:
            x_source, _, d = x[0]
            ax, bx = d["a"], d["b"]
        return z_source, x_source, az, bz, ax, bx

    def _parse_message_a(self, message):
        z = filter_message(message, "fwd")
        x = filter_message(message, "bwd")
        az = tau_z = z_source = ax = x_source = None
        if self.n_prev == 1:
            z_source, _, d = z[0]
            az, tau_z = d["a"], d.get("tau")
        if self.n_next == 1:
            x_source, _, d = x[0]
            ax = d["a"]
        return z_source, x_source, az, ax, tau_z

    def forward_message(self, message):
        if self.n_next == 0:
            return []
        z_source, x_source, az, bz, ax, bx = self._parse_message_ab(message)
        if self.n_prev == 0:
            ax_new, bx_new = self.compute_forward_message(ax, bx)
        else:
            ax_new, bx_new = self.compute_forward_message(az, bz, ax, bx)
        return [(self, x_source, dict(a=ax_new, b=bx_new, direction="fwd"))]

    def backward_message(self, message):
        if self.n_prev == 0:
            return []
        z_source, x_source, az, bz, ax, bx = self._parse_message_ab(message)
        if self.n_next == 0:
            az_new, bz_new = self.compute_backward_message(az, bz)
        else:
            az_new, bz_new = self.compute_backward_message(az, bz, ax, bx)
        return [(self, z_source, dict(a=az_new, b=bz_new, direction="bwd"))]

    def forward_state_evolution(self, message):
        if self.n_next == 0:
            return []
        z_source, x_source, az, ax, tau_z = self._parse_message_a(message)
        if self.n_prev == 0:
            ax_new = self.compute_forward_state_evolution(ax)
        else:
            ax_new = self.compute_forward_state_evolution(az, ax, tau_z)
        return [(self, x_source, dict(a=ax_new, direction="fwd"))]

    def backward_state_evolution(self, message):
        if self.n_prev == 0:
            return []
        z_source, x_source, az, ax, tau_z = self._parse_message_a(message)
        if self.n_next == 0:
            az_new = self.compute_backward_state_evolution(az, tau_z)
        else:
            az_new = self.compute_backward_state_evolution(az, ax, tau_z)
        return [(self, z_source, dict(a=az_new, direction="bwd"))]


__all__ = ["Variable", "Factor", "VecArg", "ScalArg", "filter_message", "inv", "AMIN", "AMAX", "_lib"]
