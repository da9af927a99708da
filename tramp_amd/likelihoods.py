"""Likelihoods (leaf factors, y observed): host mirror of tramp.likelihoods.

tramp/likelihoods/{base_likelihood,gaussian_likelihood,sgn_likelihood,abs_likelihood}.py.
``y`` is uploaded once at construction and stays resident.
"""
import numpy as np

from . import _lib
from .base import Factor, VecArg, ScalArg, out_vec, ret_vec, ret_scal
from .device import DeviceArray


class Likelihood(Factor):
    "tramp/likelihoods/base_likelihood.py:6-118"
    n_next = 0
    n_prev = 1
    _kernel_extra = ()

    def get_size(self, y):
        if y is None:
            return None
        return y.shape[0] if len(y.shape) == 1 else y.shape

    def _set_y(self, y):
        self.y = y
        self._y_dev = None
        if y is None:
            return
        if isinstance(y, DeviceArray):
            self._y_dev = y if y.ndim == 2 else y.reshape(1, y.shape[0])
        else:
            self._y_dev = self.ctx.array(np.atleast_2d(np.asarray(y, dtype=self.dtype)), self.dtype)

    def bind_shape(self, shape):
        """The scheduler knows the shape of the observed variable: an observation of exactly that shape is ONE instance (an
        image y of shape (h, w) is not h instances of w components), (B,) + shape is an ensemble of B.  Re-views the device
        copy as [instances, components]."""
        yd = self._y_dev
        if yd is None or shape is None:
            return
        n = int(np.prod(shape))
        if n == 0 or yd.size % n:
            raise ValueError(f"observation of shape {tuple(np.shape(self.y))} does not match the variable's shape {tuple(shape)}")
        if yd.shape != (yd.size // n, n):
            self._y_dev = yd.reshape(yd.size // n, n)
            self._rebind()

    def _rebind(self):
        "hook: state derived from the [instances, components] view of y"

    def _y_for(self, b: VecArg, y=None):
        yd = self._y_dev
        if y is not None and y is not self.y:
            yd = VecArg(self.ctx, y, self.dtype).dev
        if yd is None:
            raise ValueError("likelihood has no observation y")
        if yd.shape != (b.batch, b.n):
            raise ValueError(f"y has shape {yd.shape}, message has {(b.batch, b.n)}")
        return yd

    def _message(self, az, bz, y=None, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None,
                 want_posterior=False):
        ctx = self.ctx
        b = VecArg(ctx, bz, self.dtype)
        a = ScalArg(ctx, az, b.batch)
        yd = self._y_for(b, y)
        a_new = out_a if out_a is not None else ctx.empty((b.batch,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, b, self.dtype)
        r = out_vec(ctx, b, self.dtype) if want_posterior else None
        v = ctx.empty((b.batch,), np.float64) if want_posterior else None
        ctx._call(self._kernel_name, self._dt, b.n, b.batch, *self._kernel_extra, a.dev.ptr, b.dev.ptr, yd.ptr,
                  float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr,
                  r.ptr if r is not None else None, v.ptr if v is not None else None)
        return a, b, a_new, b_new, r, v

    _lik_hint_kind = None

    def lik_hint(self, batch, n):
        """(kind, y device array) for tramp_lik_hint: the producer of this likelihood's input message can leave
        sum v_i (csrc/prior_hint.h), or None"""
        yd = self._y_dev
        if self._lik_hint_kind is None or not self.isotropic or yd is None or yd.shape != (batch, n):
            return None
        return self._lik_hint_kind, yd

    def compute_backward_message(self, az, bz, **kw):
        "Likelihood.compute_backward_message (base_likelihood.py:25-28)"
        if not self.isotropic:
            a_new, b_new, _, _ = self._diag(az, bz, self.y)
            return a_new, b_new
        a, b, a_new, b_new, _, _ = self._message(az, bz, **kw)
        return ret_scal(a_new, a, b), ret_vec(b_new, b)

    def compute_backward_posterior(self, az, bz, y):
        if not self.isotropic:
            _, _, r, v = self._diag(az, bz, y)
            return r, v
        a, b, _, _, r, v = self._message(az, bz, y=y, want_posterior=True)
        return ret_vec(r, b), ret_scal(v, a, b)

    _logz_kind = None
    _logz_params = ()

    def compute_log_partition(self, az, bz, y=None):
        "mean over the components of the likelihood's log-partition (e.g. sgn_likelihood.py:39-41)"
        if self._logz_kind is None:
            if not hasattr(self, "_regions"):
                raise NotImplementedError(f"{type(self).__name__}.compute_log_partition is not on the HIP path")
            # the observed piecewise-linear outputs: the reference's loop over its region table
            # (piecewise_linear_likelihood.py:52-61, 216-219)
            yy = self.y if y is None else y
            if isinstance(yy, DeviceArray) or np.ndim(yy) == 0:
                yy = self._y_for(VecArg(self.ctx, bz, self.dtype), None)
            return self._logz_regions(1, self._regions(), az, bz, yy, mean=True)
        b = VecArg(self.ctx, bz, self.dtype)
        yd = self._y_for(b, y)
        return self._logz(self._logz_kind, az, bz, v1=yd, params=self._logz_params)

    def compute_backward_state_evolution(self, az, tau_z):
        "base_likelihood.py:73-76"
        vz = self.compute_backward_error(az, tau_z)
        return self.compute_a_new(vz, az)

    def compute_backward_overlap(self, az, tau_z):
        return tau_z - self.compute_backward_error(az, tau_z)


class GaussianLikelihood(Likelihood):
    _logz_kind = _lib.LOGZ_GAUSS_LIK
    "tramp/likelihoods/gaussian_likelihood.py:7-132 (K6: constant message)"

    def __init__(self, y, var=1, y_name="y", isotropic=True, dtype=np.float64, ctx=None):
        self.y_name = y_name
        self.size = self.get_size(y)
        self.var = var
        self._logz_params = (float(var),)
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self.sigma = np.sqrt(var)
        self.a = 1 / var
        self._set_y(y)
        self._b_dev = None      # y / var, formed once on the device (gaussian_likelihood.py:17)

    _diag_which = _lib.DIAG_GAUSS_LIK

    def _diag_params(self):
        return (self.var,)

    def sample(self, X):
        return X + self.sigma * np.random.standard_normal(X.shape)

    def compute_backward_posterior(self, az, bz, y):
        """gaussian_likelihood.py:43-49: r = (bz + y/var) / (az + 1/var), v = 1 / (az + 1/var): kernel K11 on the
        incoming message and the likelihood's own constant one."""
        if not self.isotropic:
            _, _, r, v = self._diag(az, bz, y)
            return r, v
        ctx = self.ctx
        b = VecArg(ctx, bz, self.dtype)
        a = ScalArg(ctx, az, b.batch)
        yd = self._y_for(b, y)
        a0, b0 = ctx.empty((b.batch,), np.float64), out_vec(ctx, b, self.dtype)
        ctx._call("tramp_gauss_lik_bwd", self._dt, b.n, b.batch, yd, float(self.var), a0, b0)
        r, v = out_vec(ctx, b, self.dtype), ctx.empty((b.batch,), np.float64)
        ctx._call("tramp_variable_posterior", self._dt, b.n, b.batch, a.dev, b.dev, a0, b0, None, r, v, None, None)
        return ret_vec(r, b), ret_scal(v, a, b)

    def compute_backward_message(self, az, bz, out_a=None, out_b=None, **_):
        "gaussian_likelihood.py:68-71"
        if not self.isotropic:
            return self.a, self.a * np.asarray(self.y, dtype=self.dtype)
        ctx = self.ctx
        b = VecArg(ctx, bz, self.dtype)
        a = ScalArg(ctx, az, b.batch)
        yd = self._y_for(b)
        a_new = out_a if out_a is not None else ctx.empty((b.batch,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, b, self.dtype)
        ctx._call("tramp_gauss_lik_bwd", self._dt, b.n, b.batch, yd.ptr, float(self.var), a_new.ptr, b_new.ptr)
        return ret_scal(a_new, a, b), ret_vec(b_new, b)

    def compute_backward_error(self, az, tau_z):
        "gaussian_likelihood.py:58-61"
        return 1 / (az + self.a)

    def compute_backward_state_evolution(self, az, tau_z):
        "gaussian_likelihood.py:73-75"
        return self.a


class SgnLikelihood(Likelihood):
    _logz_kind = _lib.LOGZ_SGN_LIK
    "tramp/likelihoods/sgn_likelihood.py:9-98 (K4)"
    _kernel_name = "tramp_sgn_lik_bwd"
    _diag_which = _lib.DIAG_SGN_LIK
    _lik_hint_kind = 3          # csrc/prior_hint.h PRIOR_HINT_SGN_LIK

    def __init__(self, y, y_name="y", isotropic=True, dtype=np.float64, ctx=None):
        self.y_name = y_name
        self.size = self.get_size(y)
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self._set_y(y)

    def repack(self):
        """Pack the observation again from the device buffer it lives in.  The packed copy is made ONCE, when y is set
        (the reference's y is static: sgn_likelihood.py:14-19); a caller that passed a DeviceArray and overwrites it in
        place must call this, or the large-batch updates keep reading the old signs."""
        self._pack()

    def _set_y(self, y):
        """the observation is static and in {-1, +1}: besides y itself keep it packed one bit per component, which is
        what batches of large instances read (tramp_sgn_lik_bwd_bits: 2T + n/8 bytes moved instead of 3T).  FROZEN at
        this point: see repack()."""
        super()._set_y(y)
        self._pack()

    def _rebind(self):
        self._pack()

    def _pack(self):
        self._bits = None
        yd = self._y_dev
        if yd is None or not self.isotropic or yd.shape[0] > 65535 or yd.shape[1] == 0:
            return
        import ctypes as C
        B, n = yd.shape
        bits = self.ctx.empty((B, ((n + 63) // 64) * 8), np.uint8)
        bad = C.c_ulonglong()
        self.ctx._call("tramp_sgn_pack", self._dt, n, B, yd.ptr, bits.ptr, C.byref(bad))
        if bad.value == 0:
            self._bits = bits

    def _message(self, az, bz, y=None, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None,
                 want_posterior=False):
        if want_posterior or self._bits is None or (y is not None and y is not self.y):
            return super()._message(az, bz, y=y, out_a=out_a, out_b=out_b, damping=damping, old_a=old_a,
                                    old_b=old_b, want_posterior=want_posterior)
        ctx = self.ctx
        b = VecArg(ctx, bz, self.dtype)
        a = ScalArg(ctx, az, b.batch)
        yd = self._y_for(b, None)
        a_new = out_a if out_a is not None else ctx.empty((b.batch,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, b, self.dtype)
        ctx._call("tramp_sgn_lik_bwd_bits", self._dt, b.n, b.batch, a.dev.ptr, b.dev.ptr, yd.ptr, self._bits.ptr,
                  float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr, None, None)
        return a, b, a_new, b_new, None, None

    def lik_hint(self, batch, n):
        """no producer-fused moments where the packed resident update runs: it moves 2T, the fused form 3T"""
        if self._bits is not None and self.ctx.resident_covers(self.dtype, n, batch):
            return None
        return super().lik_hint(batch, n)

    def sample(self, X):
        return np.sign(X)

    def compute_backward_error(self, az, tau_z):
        "base_likelihood.py:78-81 with beliefs_measure sgn_likelihood.py:79-92 (K13)"
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        tz = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        if np.any(az_h - 1 / tz <= 0):
            raise AssertionError("az must be greater than 1/ tau_z")     # sgn_likelihood.py:80-81
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_sgn_lik", az_h.shape[0], az_h.ctypes.data, tz.ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(az) == 0 else v


class AbsLikelihood(Likelihood):
    _logz_kind = _lib.LOGZ_ABS_LIK
    "tramp/likelihoods/abs_likelihood.py:8-72 (K5)"
    _kernel_name = "tramp_abs_lik_bwd"
    _diag_which = _lib.DIAG_ABS_LIK
    _lik_hint_kind = 4          # PRIOR_HINT_ABS_LIK

    def __init__(self, y, y_name="y", isotropic=True, dtype=np.float64, ctx=None):
        self.y_name = y_name
        self.size = self.get_size(y)
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self._set_y(y)

    def sample(self, X):
        return np.abs(X)

    def compute_backward_error(self, az, tau_z):
        "beliefs_measure abs_likelihood.py:60-69 (2-D quadrature, K14 family)"
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        tz = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        if np.any(az_h - 1 / tz <= 0):
            raise AssertionError("az must be greater than 1/tau_z")
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_abs_lik", az_h.shape[0], az_h.ctypes.data, tz.ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(az) == 0 else v


class ModulusLikelihood(Likelihood):
    """Modulus likelihood y = |z| for complex z: the output factor of phase retrieval.

    tramp/likelihoods/modulus_likelihood.py:36-131.  A complex array is its planar real form Z[0] = Re z, Z[1] = Im z
    (tramp/utils/misc.py:13-27): host messages are (2, n) or (batch, 2, n), device messages [batch, 2 n]; y is (n,) or
    (batch, n).  Kernel: csrc/modulus.hip (Bessel ratio I1 / I0 in double).
    """

    def bind_shape(self, shape):
        "the observed variable is planar complex (2, n): y = |z| has the shape of ONE plane"
        super().bind_shape(None if shape is None else tuple(shape[1:]))

    def __init__(self, y, y_name="y", isotropic=True, dtype=np.float64, ctx=None):
        if not isotropic:
            raise NotImplementedError("diagonal beliefs for ModulusLikelihood are outside the HIP path")
        self.y_name = y_name
        self.size = self.get_size(y)
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self._set_y(y)

    def sample(self, Z):
        "We assume Z[0] = Z.real and Z[1] = Z.imag (modulus_likelihood.py:58-61)"
        Z = np.asarray(Z)
        return np.absolute(Z[0] + 1j * Z[1])

    def _planar(self, bz):
        "(device [batch, 2 n], the host shape to give results back in or None)"
        if isinstance(bz, DeviceArray):
            return (bz if bz.ndim == 2 else bz.reshape(1, bz.shape[0])), None
        h = np.asarray(bz, dtype=self.dtype)
        if h.ndim not in (2, 3) or h.shape[-2] != 2:
            raise ValueError(f"expected a planar complex message (2, n) or (batch, 2, n), got {h.shape}")
        return self.ctx.array(np.ascontiguousarray(h.reshape(-1, 2 * h.shape[-1])), self.dtype), h.shape

    def _run(self, az, bz, y=None, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None, want_posterior=False):
        ctx = self.ctx
        bd, hshape = self._planar(bz)
        batch, n2 = bd.shape
        n = n2 // 2
        a = ScalArg(ctx, az, batch)
        yd = self._y_dev
        if y is not None and y is not self.y:
            yd = y if isinstance(y, DeviceArray) else ctx.array(np.atleast_2d(np.asarray(y, dtype=self.dtype)), self.dtype)
        if yd is None:
            raise ValueError("likelihood has no observation y")
        if tuple(yd.shape) != (batch, n):
            raise ValueError(f"y has shape {tuple(yd.shape)}, message has {(batch, 2, n)}")
        a_new = out_a if out_a is not None else ctx.empty((batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((batch, n2), self.dtype)
        r = ctx.empty((batch, n2), self.dtype) if want_posterior else None
        v = ctx.empty((batch,), np.float64) if want_posterior else None
        ctx._call("tramp_modulus_lik_bwd", self._dt, n, batch, a.dev.ptr, bd.ptr, yd.ptr, float(damping or 0.0),
                  old_a.ptr if old_a is not None else None, old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr,
                  r.ptr if r is not None else None, v.ptr if v is not None else None)
        return a, hshape, a_new, b_new, r, v

    @staticmethod
    def _ret_vec(dev, hshape):
        return dev if hshape is None else dev.get().reshape(hshape)

    @staticmethod
    def _ret_scal(dev, a, hshape):
        if hshape is None or not a.was_host:
            return dev
        h = dev.get()
        return float(h[0]) if (a.scalar_like and len(hshape) == 2) else h

    def compute_backward_posterior(self, az, bz, y):
        "modulus_likelihood.py:82-93"
        a, hs, _, _, r, v = self._run(az, bz, y=y, want_posterior=True)
        return self._ret_vec(r, hs), self._ret_scal(v, a, hs)

    def compute_backward_message(self, az, bz, **kw):
        "Likelihood.compute_backward_message (base_likelihood.py:25-28)"
        a, hs, a_new, b_new, _, _ = self._run(az, bz, **kw)
        return self._ret_scal(a_new, a, hs), self._ret_vec(b_new, hs)

    def compute_log_partition(self, az, bz, y=None):
        "modulus_likelihood.py:125-129, as found: |.| of each real coordinate, A.mean() / 2"
        ctx = self.ctx
        bd, hs = self._planar(bz)
        batch, n2 = bd.shape
        a = ScalArg(ctx, az, batch)
        yd = self._y_dev
        if y is not None and y is not self.y:
            yd = y if isinstance(y, DeviceArray) else ctx.array(np.atleast_2d(np.asarray(y, dtype=self.dtype)), self.dtype)
        out = ctx.empty((batch,), np.float64)
        ctx._call("tramp_modulus_log_partition", self._dt, n2 // 2, batch, a.dev.ptr, bd.ptr, yd.ptr, out.ptr)
        res = out.get() / (2.0 * n2)
        return float(res[0]) if (hs is not None and a.was_host and a.scalar_like and len(hs) == 2) else res

    def compute_backward_error(self, az, tau_z):
        "beliefs_measure modulus_likelihood.py:101-120 (2-D Bessel-weighted average, K14 family: tramp_se_modulus_lik)"
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        tz = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_modulus_lik", az_h.shape[0], az_h.ctypes.data, tz.ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(az) == 0 else v


class ReluLikelihood(Likelihood):
    """y = relu(z) observed (tramp/likelihoods/piecewise_linear_likelihood.py:279-284): y > 0 pins
    z = y, y == 0 leaves a truncated normal on (-inf, 0]."""
    _kernel_name = "tramp_relu_lik_bwd"
    _slope = 0.0
    _diag_which = _lib.DIAG_RELU_LIK

    def _diag_params(self):
        return (self._slope,)

    def __init__(self, y, y_name="y", isotropic=True, dtype=np.float64, ctx=None):
        self.y_name = y_name
        self.size = self.get_size(y)
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self._kernel_extra = (float(self._slope),)
        self._set_y(y)

    def sample(self, X):
        return np.where(X >= 0, X, self._slope * X)

    def _regions(self):
        "[pos, neg] rows (zmin, zmax, slope, x0) (piecewise_linear_likelihood.py:279-291)"
        return [(0.0, np.inf, 1.0, 0.0), (-np.inf, 0.0, float(self._slope), 0.0)]

    def compute_backward_error(self, az, tau_z):
        "beliefs_measure (piecewise_linear_likelihood.py:120-146): only flat regions carry variance"
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        tz = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        if np.any(az_h - 1 / tz <= 0):
            raise AssertionError("az must be greater than 1/ tau_z")
        if self._slope != 0:
            return 0.0 if np.ndim(az) == 0 else np.zeros_like(az_h)
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_relu_lik", az_h.shape[0], az_h.ctypes.data, tz.ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(az) == 0 else v


class LeakyReluLikelihood(ReluLikelihood):
    "piecewise_linear_likelihood.py:286-291"

    def __init__(self, slope, y, y_name="y", isotropic=True, dtype=np.float64, ctx=None):
        self.slope = slope
        self._slope = float(slope)
        super().__init__(y, y_name, isotropic, dtype, ctx)


class ThreeRegionLikelihood(Likelihood):
    """Observed three-region output (the likelihood twins of
    HardTanh / HardSigmoid / SymmetricDoor): tramp/likelihoods/piecewise_linear_likelihood.py
    :156-218 over the regions pos [z_hi, inf), mid [z_lo, z_hi], neg (-inf, z_lo]."""
    _kernel_name = "tramp_pwl3_lik_bwd"
    _z = (-1.0, 1.0)
    _slopes = (0.0, 1.0, 0.0)     # pos, mid, neg
    _x0 = (1.0, 0.0, -1.0)
    _diag_which = _lib.DIAG_PWL3_LIK

    def _diag_params(self):
        return tuple(self._z) + tuple(self._slopes) + tuple(self._x0)

    def __init__(self, y, y_name="y", isotropic=True, dtype=np.float64, ctx=None):
        self.y_name = y_name
        self.size = self.get_size(y)
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self._slopes_h = np.array(self._slopes, dtype=np.float64)
        self._x0_h = np.array(self._x0, dtype=np.float64)
        self._kernel_extra = (float(self._z[0]), float(self._z[1]), self._slopes_h.ctypes.data,
                              self._x0_h.ctypes.data)
        self._set_y(y)

    def _regions(self):
        "[pos, mid, neg] rows (zmin, zmax, slope, x0)"
        zlo, zhi = self._z
        (sp, sm, sn), (xp, xm, xn) = self._slopes, self._x0
        return [(zhi, np.inf, sp, xp), (zlo, zhi, sm, xm), (-np.inf, zlo, sn, xn)]

    def sample(self, Z):
        "sum of the regions' outputs: x0 + slope z on [zmin, zmax), zero elsewhere (:22-25, 168-170)"
        zlo, zhi = self._z
        (sp, sm, sn), (xp, xm, xn) = self._slopes, self._x0
        return np.where(Z >= zhi, xp + sp * Z, np.where(Z >= zlo, xm + sm * Z, xn + sn * Z))

    def compute_backward_error(self, az, tau_z):
        "beliefs_measure summed over the regions (:120-146, 210-214); sloped regions carry no variance"
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        tz = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        if np.any(az_h - 1 / tz <= 0):
            raise AssertionError("az must be greater than 1/ tau_z")
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_pwl3_lik", az_h.shape[0], float(self._z[0]), float(self._z[1]),
                       self._slopes_h.ctypes.data, self._x0_h.ctypes.data, az_h.ctypes.data, tz.ctypes.data,
                       v.ctypes.data)
        return float(v[0]) if np.ndim(az) == 0 else v


class HardTanhLikelihood(ThreeRegionLikelihood):
    "piecewise_linear_likelihood.py:293-299"


class HardSigmoidLikelihood(ThreeRegionLikelihood):
    "piecewise_linear_likelihood.py:302-309 (l = 3 here; the channel of the same name uses 2.5)"
    _z = (-3.0, 3.0)
    _slopes = (0.0, 1.0 / 6.0, 0.0)
    _x0 = (1.0, 0.5, 0.0)


class SymmetricDoorLikelihood(ThreeRegionLikelihood):
    "piecewise_linear_likelihood.py:312-320"

    def __init__(self, width, y, y_name="y", isotropic=True, dtype=np.float64, ctx=None):
        self.width = width
        self._z = (-float(width), float(width))
        self._slopes = (0.0, 0.0, 0.0)
        self._x0 = (1.0, -1.0, 1.0)
        super().__init__(y, y_name, isotropic, dtype, ctx)


class AsymmetricAbsLikelihood(ThreeRegionLikelihood):
    """y = |z| observed with the kink moved to z = shift (piecewise_linear_likelihood.py:260-265): regions
    pos (shift, inf) with y = z and neg (-inf, shift) with y = -z.  Both pin z: y > |shift| is compatible with
    z = +-y (weights softmax(+-bz y): the abs likelihood's tanh), -shift < y <= shift only with z = -y, y <= -shift
    with neither (NaN, as the reference's softmax of two -inf).

    The reference itself returns NaN for EVERY input: LinearRegionLikelihood.log_partitions (:52-61) subtracts
    np.log(self.slope) and the descending branch has slope = -1.  The Jacobian of y = x0 + slope z is |slope|;
    that is what the kernel takes (log|-1| = 0).  ``reference_nan=True`` reproduces the reference's outputs bit for
    bit instead.  On the device it is the three-region likelihood kernel with an empty middle
    region (z_lo = z_hi = shift)."""
    _logz_kind = None

    def __init__(self, y, y_name="y", isotropic=True, shift=1e-4, reference_nan=False, dtype=np.float64, ctx=None):
        self.shift = shift
        self.reference_nan = bool(reference_nan)
        self._z = (float(shift), float(shift))
        self._slopes = (1.0, 1.0, -1.0)        # pos, (empty) mid, neg
        self._x0 = (0.0, 0.0, 0.0)
        super().__init__(y, y_name, isotropic, dtype, ctx)
        self.name = "a-abs"

    def _regions(self):
        s = float(self.shift)
        return [(s, np.inf, 1.0, 0.0), (-np.inf, s, -1.0, 0.0)]

    def sample(self, Z):
        return np.where(Z >= self.shift, Z, -Z)

    def _nan_like(self, bz):
        b = np.asarray(bz.get() if isinstance(bz, DeviceArray) else bz, dtype=np.float64)
        return np.full(b.shape, np.nan)

    def compute_backward_message(self, az, bz, **kw):
        if self.reference_nan:
            nb = self._nan_like(bz)
            return (np.nan if nb.ndim == 1 else np.full(nb.shape[0], np.nan)), nb
        return super().compute_backward_message(az, bz, **kw)

    def compute_backward_posterior(self, az, bz, y):
        if self.reference_nan:
            nb = self._nan_like(bz)
            return nb, (np.nan if nb.ndim == 1 or not self.isotropic else np.full(nb.shape[0], np.nan))
        return super().compute_backward_posterior(az, bz, y)

    def compute_log_partition(self, az, bz, y=None):
        if self.reference_nan:
            return float("nan")
        return super().compute_log_partition(az, bz, y)

    def compute_backward_error(self, az, tau_z):
        """E[v] over z ~ N(0, tau_z), bz = mz z + sqrt(mz) xi (beliefs_measure :120-146): the posterior is the abs
        likelihood's wherever both branches are compatible (|z| > |shift|) and a point mass elsewhere, so the error
        is tramp_se_abs_lik minus the slab |z| <= |shift|, which a 64 x 64 Gauss rule on the host integrates."""
        if self.reference_nan:
            return float("nan") if np.ndim(az) == 0 else np.full(np.shape(az), np.nan)
        az_h = np.atleast_1d(np.asarray(az, dtype=np.float64))
        tz = np.broadcast_to(np.asarray(tau_z, dtype=np.float64), az_h.shape).copy()
        if np.any(az_h - 1 / tz <= 0):
            raise AssertionError("az must be greater than 1/tau_z")
        v = np.empty_like(az_h)
        self.ctx._call("tramp_se_abs_lik", az_h.shape[0], az_h.ctypes.data, tz.ctypes.data, v.ctypes.data)
        s = abs(float(self.shift))
        if s > 0:
            zg, zw = np.polynomial.legendre.leggauss(64)
            xg, xw = np.polynomial.hermite_e.hermegauss(64)
            z = s * zg[:, None]                                   # the slab, both signs
            for i, (a_, t_) in enumerate(zip(az_h, tz)):
                mz = a_ - 1 / t_
                bz = mz * z + np.sqrt(mz) * xg[None, :]
                f = (z**2) / np.cosh(bz * np.abs(z))**2 * np.exp(-0.5 * z**2 / t_) / np.sqrt(2 * np.pi * t_)
                v[i] -= s * float(zw @ (f @ (xw / np.sqrt(2 * np.pi))))
        return float(v[0]) if np.ndim(az) == 0 else v


LIKELIHOOD_CLASSES = {
    "a-abs": AsymmetricAbsLikelihood,
    "relu": ReluLikelihood,
    "l-relu": LeakyReluLikelihood,
    "gaussian": GaussianLikelihood,
    "abs": AbsLikelihood,
    "sgn": SgnLikelihood,
    "h-tanh": HardTanhLikelihood,
    "h-sigm": HardSigmoidLikelihood,
    "door": SymmetricDoorLikelihood,
    "modulus": ModulusLikelihood,
}


def get_likelihood(y, likelihood_type, **kwargs):
    "tramp/likelihoods/__init__.py:25-27"
    return LIKELIHOOD_CLASSES[likelihood_type](y=y, **kwargs)
