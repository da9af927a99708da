"""Priors (root factors): host mirror of tramp.priors for the hot-path classes.

Same class names, constructor arguments and method signatures as
tramp/priors/{base_prior,gauss_bernoulli_prior,map_L1_norm_prior,gaussian_prior}.py;
the method bodies launch HIP kernels through the C ABI.  Extra keyword-only
arguments (``out_a``, ``out_b``, ``damping``, ``old_a``, ``old_b``) serve the
device-resident scheduler and have no counterpart in the reference.
"""
import numpy as np

from . import _lib
from .base import Factor, VecArg, ScalArg, out_vec, ret_vec, ret_scal
from .device import DeviceArray


class Prior(Factor):
    "tramp/priors/base_prior.py:6-107"
    n_next = 1
    n_prev = 0

    def _message(self, kernel, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None,
                 want_posterior=False):
        ctx = self.ctx
        b = VecArg(ctx, bx, self.dtype)
        a = ScalArg(ctx, ax, b.batch)
        a_new = out_a if out_a is not None else ctx.empty((b.batch,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, b, self.dtype)
        r = out_vec(ctx, b, self.dtype) if want_posterior else None
        v = ctx.empty((b.batch,), np.float64) if want_posterior else None
        kernel(b, a, float(damping or 0.0),
               old_a.ptr if old_a is not None else None, old_b.ptr if old_b is not None else None,
               a_new, b_new, r, v)
        return a, b, a_new, b_new, r, v

    def compute_forward_message(self, ax, bx, **kw):
        "Prior.compute_forward_message (base_prior.py:13-16)"
        if not self.isotropic:
            a_new, b_new, _, _ = self._diag(ax, bx)
            return a_new, b_new
        a, b, a_new, b_new, _, _ = self._message(self._kernel, ax, bx, **kw)
        return ret_scal(a_new, a, b), ret_vec(b_new, b)

    def compute_forward_posterior(self, ax, bx):
        if not self.isotropic:
            _, _, r, v = self._diag(ax, bx)
            return r, v
        a, b, _, _, r, v = self._message(self._kernel, ax, bx, want_posterior=True)
        return ret_vec(r, b), ret_scal(v, a, b)

    def compute_forward_state_evolution(self, ax):
        "base_prior.py:61-64"
        vx = self.compute_forward_error(ax)
        return self.compute_a_new(vx, ax)

    def compute_forward_overlap(self, ax):
        return self.second_moment() - self.compute_forward_error(ax)

    def _logz2(self, which, params, ax, bx):
        "mean over the components of the prior's log-partition (tramp_prior2_log_partition returns the sums)"
        import ctypes as C
        ctx = self.ctx
        if not self.isotropic and np.ndim(ax) > 0 and not isinstance(bx, DeviceArray):
            # diagonal beliefs: n instances of one component each, then the reference's A.mean()
            vh = np.asarray(bx, dtype=self.dtype).reshape(-1, 1)
            ah = np.broadcast_to(np.asarray(ax, dtype=np.float64).ravel(), (vh.shape[0],)).copy()
            iso, self.isotropic = self.isotropic, True
            try:
                per = np.atleast_1d(self._logz2(which, params, ah, vh))
            finally:
                self.isotropic = iso
            return float(per.mean())
        v = VecArg(ctx, bx, self.dtype)
        a = ScalArg(ctx, ax, v.batch)
        par = [float(x) for x in params]
        arr = (C.c_double * len(par))(*par)
        out = ctx.empty((v.batch,), np.float64)
        ctx._call("tramp_prior2_log_partition", self._dt, int(which), v.n, v.batch, arr, len(par), a.dev.ptr, v.dev.ptr,
                  out.ptr)
        res = out.get() / v.n
        if a.was_host and a.scalar_like and v.was_host and len(v.host_shape) == 1:
            return float(res[0])
        return res


class GaussBernoulliPrior(Prior):
    r"""Gauss-Bernoulli prior p(x) = (1-rho) delta(x) + rho N(x | mean, var).

    tramp/priors/gauss_bernoulli_prior.py:8-126 (kernel K1).
    """

    def __init__(self, size, rho=0.5, mean=0, var=1, isotropic=True, dtype=np.float64, ctx=None):
        self.size = size
        self.rho = rho
        self.mean = mean
        self.var = var
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self.sigma = np.sqrt(var)
        self.a = 1 / var
        self.b = mean / var
        # normal.A(a, b) - log(rho / (1 - rho))   (gauss_bernoulli_prior.py:36)
        self.eta = 0.5 * (self.b**2 / self.a + np.log(2 * np.pi / self.a)) - np.log(rho / (1 - rho))

    def sample(self):
        "gauss_bernoulli_prior.py:38-42"
        X_gauss = self.mean + self.sigma * np.random.standard_normal(self.size)
        X_bernoulli = np.random.binomial(n=1, size=self.size, p=self.rho)
        return X_gauss * X_bernoulli

    def second_moment(self):
        return self.rho * (self.mean**2 + self.var)

    def compute_log_partition(self, ax, bx):
        "gauss_bernoulli_prior.py:79-83"
        return self._logz(_lib.LOGZ_GB_PRIOR, ax, bx, params=(self.rho, self.mean, self.var))

    def _kernel(self, b, a, damping, old_a, old_b, a_new, b_new, r, v):
        self.ctx._call("tramp_gb_prior_fwd", self._dt, b.n, b.batch, a.dev.ptr, b.dev.ptr,
                       float(self.rho), float(self.mean), float(self.var), damping, old_a, old_b,
                       a_new.ptr, b_new.ptr, r.ptr if r is not None else None,
                       v.ptr if v is not None else None)

    _diag_which = _lib.DIAG_GB

    def _diag_params(self):
        return (self.rho, self.mean, self.var)

    def moment_hint(self):
        "(kind, p0, p1, p2) for tramp_prior_hint: the producer of this prior's input message can leave sum v_i"
        return (1, float(self.rho), float(self.mean), float(self.var)) if self.isotropic else None

    def compute_forward_error(self, ax):
        """SE error E_b[v(ax, b)] (base_prior.py:66-69, beliefs_measure
        gauss_bernoulli_prior.py:116-122): quadrature kernel K13."""
        ax_h = np.atleast_1d(np.asarray(ax, dtype=np.float64))
        n = ax_h.shape[0]
        par = [np.full(n, float(p)) for p in (self.rho, self.mean, self.var)]
        v = np.empty(n)
        self.ctx._call("tramp_se_gb_prior", n, ax_h.ctypes.data, par[0].ctypes.data, par[1].ctypes.data,
                       par[2].ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(ax) == 0 else v


class MAP_L1NormPrior(Prior):
    r"""MAP prior of the penalty gamma ||x||_1: soft-threshold denoiser.

    tramp/priors/map_L1_norm_prior.py:20-88 (kernel K2).
    """

    def __init__(self, size, gamma=1, isotropic=True, dtype=np.float64, ctx=None):
        self.size = size
        self.gamma = gamma
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)

    def sample(self):
        return np.random.laplace(size=self.size, scale=1 / self.gamma)

    def second_moment(self):
        raise NotImplementedError

    def compute_log_partition(self, ax, bx):
        "map_L1_norm_prior.py:73-76"
        return self._logz(_lib.LOGZ_L1_PRIOR, ax, bx, params=(self.gamma,))

    def _kernel(self, b, a, damping, old_a, old_b, a_new, b_new, r, v, count=None, mask=None):
        self.ctx._call("tramp_l1_prior_fwd", self._dt, b.n, b.batch, a.dev.ptr, b.dev.ptr,
                       float(self.gamma), damping, old_a, old_b, a_new.ptr, b_new.ptr,
                       r.ptr if r is not None else None, v.ptr if v is not None else None,
                       count.ptr if count is not None else None, mask.ptr if mask is not None else None)

    def support(self, ax, bx):
        "Exact support mask |bx| > gamma and its count (v_soft_threshold, map_L1_norm_prior.py:16-17)."
        ctx = self.ctx
        b = VecArg(ctx, bx, self.dtype)
        a = ScalArg(ctx, ax, b.batch)
        count = ctx.empty((b.batch,), np.uint64)
        mask = ctx.empty((b.batch, b.n), np.uint8)
        self._kernel(b, a, 0.0, None, None, ctx.empty((b.batch,), np.float64),
                     out_vec(ctx, b, self.dtype), None, None, count, mask)
        return mask.get().reshape(b.host_shape).astype(bool), count.get()

    _diag_which = _lib.DIAG_L1

    def _diag_params(self):
        return (self.gamma,)

    def moment_hint(self):
        return (2, float(self.gamma), 0.0, 0.0) if self.isotropic else None

    def compute_forward_error(self, ax):
        raise NotImplementedError      # as the reference (map_L1_norm_prior.py:84-85)


class BinaryPrior(Prior):
    r"""Binary prior p(x) = p_+ delta(x - 1) + p_- delta(x + 1).

    tramp/priors/binary_prior.py:8-62 (the prior of the sign committee machine, committee_model.py:44-54).
    """

    def __init__(self, size, p_pos=0.5, isotropic=True, dtype=np.float64, ctx=None):
        self.size = size
        self.p_pos = p_pos
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self.p_neg = 1 - p_pos
        self.b = 0.5 * np.log(self.p_pos / self.p_neg)          # :27

    _diag_which = _lib.DIAG_BINARY

    def _diag_params(self):
        return (self.b,)

    def sample(self):
        return np.random.choice([-1, +1], size=self.size, replace=True, p=[self.p_neg, self.p_pos])

    def second_moment(self):
        return 1.

    def _kernel(self, b, a, damping, old_a, old_b, a_new, b_new, r, v):
        self.ctx._call("tramp_binary_prior_fwd", self._dt, b.n, b.batch, a.dev.ptr, b.dev.ptr, float(self.b),
                       damping, old_a, old_b, a_new.ptr, b_new.ptr,
                       r.ptr if r is not None else None, v.ptr if v is not None else None)

    def compute_log_partition(self, ax, bx):
        "binary_prior.py:64-67: mean of binary.A(bx + b) - binary.A(b) - ax / 2, binary.A(b) = logaddexp(b, -b)"
        return self._logz2(2, (self.b,), ax, bx)

    def compute_forward_error(self, ax):
        "binary_prior.py:82-86 through base_prior.py:66-69, device quadrature (tramp_se_binary_prior)"
        ax_h = np.atleast_1d(np.asarray(ax, dtype=np.float64))
        n = ax_h.shape[0]
        pp, v = np.full(n, float(self.p_pos)), np.empty(n)
        self.ctx._call("tramp_se_binary_prior", n, ax_h.ctypes.data, pp.ctypes.data, v.ctypes.data)
        return float(v[0]) if np.ndim(ax) == 0 else v


class _HalfLinePrior(Prior):
    """Priors supported on x >= 0 whose tilted posterior is a half-line truncated normal (tramp/beliefs/positive.py):
    p(x) ~ 1[x >= 0] exp(-a0 x^2 / 2 + b0 x).  One kernel (tramp_halfline_prior_fwd) for both subclasses."""
    _a0 = 0.0
    _b0 = 0.0
    _diag_which = _lib.DIAG_HALFLINE

    def _diag_params(self):
        return (self._a0, self._b0)

    def _kernel(self, b, a, damping, old_a, old_b, a_new, b_new, r, v):
        self.ctx._call("tramp_halfline_prior_fwd", self._dt, b.n, b.batch, a.dev.ptr, b.dev.ptr, float(self._a0),
                       float(self._b0), damping, old_a, old_b, a_new.ptr, b_new.ptr,
                       r.ptr if r is not None else None, v.ptr if v is not None else None)

    def compute_forward_error(self, ax):
        raise NotImplementedError      # as the reference: beliefs_measure is not implemented (exponential_prior.py:73-74)


class ExponentialPrior(_HalfLinePrior):
    r"""Exponential prior p(x) = 1[x >= 0] exp(-x / mean) / mean.

    tramp/priors/exponential_prior.py:8-80: natural parameter b = -1/mean, posterior positive.r / positive.v at
    (ax, bx + b).  (The reference's ``sample`` draws with scale 1/mean, :28-30; restated as found.)
    """

    def __init__(self, size, mean=1, isotropic=True, dtype=np.float64, ctx=None):
        assert mean > 0
        self.size = size
        self.mean = mean
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self.b = -1 / mean                                       # :26
        self._a0, self._b0 = 0.0, self.b

    def sample(self):
        return np.random.exponential(scale=1 / self.mean, size=self.size)

    def second_moment(self):
        return 2 * self.mean**2

    def compute_log_partition(self, ax, bx):
        "exponential_prior.py:62-65: positive.A(ax, bx + b) - exponential.A(b), exponential.A(b) = -log(-b)"
        return self._logz2(0, (0.0, self.b, -np.log(-self.b)), ax, bx)


class PositivePrior(_HalfLinePrior):
    r"""Positive (half-normal) prior p(x) = 2 1[x >= 0] N(x | 0, 1).

    tramp/priors/positive_prior.py:8-82: natural parameters a = 1, b = 0.
    """

    def __init__(self, size, isotropic=True, dtype=np.float64, ctx=None):
        self.size = size
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self.a = 1.
        self.b = 0.
        self._a0, self._b0 = 1.0, 0.0

    def sample(self):
        return np.abs(np.random.standard_normal(self.size))     # scipy.stats.halfnorm.rvs (:27-29)

    def second_moment(self):
        return 1.

    def compute_log_partition(self, ax, bx):
        "positive_prior.py:65-69: positive.A(ax + 1, bx) - positive.A(1, 0);  positive.A(1, 0) = log(sqrt(2 pi) / 2)"
        return self._logz2(0, (1.0, 0.0, 0.5 * np.log(2 * np.pi) + np.log(0.5)), ax, bx)


class GaussianMixturePrior(Prior):
    r"""Gaussian mixture prior p(x) = sum_k p_k N(x | r_k, v_k), at most 8 components on the HIP path.

    tramp/priors/gaussian_mixture_prior.py:8-139 with tramp/beliefs/mixture.py.
    """
    _diag_which = _lib.DIAG_GM

    def __init__(self, size, probs=[0.5, 0.5], means=[-1, 1], vars=[1, 1], isotropic=True, dtype=np.float64, ctx=None):
        self.size = size
        assert len(probs) == len(means) == len(vars)
        self.K = len(probs)
        if self.K > 8:
            raise NotImplementedError("GaussianMixturePrior on the HIP path holds at most 8 components")
        self.probs = np.array(probs)
        self.means = np.array(means)
        self.vars = np.array(vars)
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self.sigmas = np.sqrt(vars)
        self.a = 1 / self.vars                                   # :36-39
        self.b = self.means / self.vars
        self.eta = np.log(self.probs) - 0.5 * (self.b**2 / self.a + np.log(2 * np.pi / self.a))

    def _tables(self):
        import ctypes as C
        return [(C.c_double * self.K)(*[float(x) for x in t]) for t in (self.probs, self.means, self.vars)]

    def _diag_params(self):
        return (self.K,) + tuple(self.probs) + tuple(self.means) + tuple(self.vars)

    def sample(self):
        shape = (self.size, self.K)
        X_gauss = self.means + self.sigmas * np.random.standard_normal(shape)
        X_cluster = np.random.multinomial(n=1, pvals=self.probs, size=self.size)
        return (X_gauss * X_cluster).sum(axis=1)

    def second_moment(self):
        return sum(p * (m**2 + v) for p, m, v in zip(self.probs, self.means, self.vars))

    def _kernel(self, b, a, damping, old_a, old_b, a_new, b_new, r, v):
        pr, me, va = self._tables()
        self.ctx._call("tramp_gm_prior_fwd", self._dt, b.n, b.batch, a.dev.ptr, b.dev.ptr, self.K, pr, me, va,
                       damping, old_a, old_b, a_new.ptr, b_new.ptr,
                       r.ptr if r is not None else None, v.ptr if v is not None else None)

    def compute_log_partition(self, ax, bx):
        "gaussian_mixture_prior.py:99-104"
        return self._logz2(1, self._diag_params(), ax, bx)

    def compute_forward_error(self, ax):
        "gaussian_mixture_prior.py:132-137 through base_prior.py:66-69, device quadrature (tramp_se_gm_prior)"
        ax_h = np.atleast_1d(np.asarray(ax, dtype=np.float64))
        pr, me, va = self._tables()
        v = np.empty(ax_h.shape[0])
        self.ctx._call("tramp_se_gm_prior", ax_h.shape[0], ax_h.ctypes.data, self.K, pr, me, va, v.ctypes.data)
        return float(v[0]) if np.ndim(ax) == 0 else v


class MAP_L21NormPrior(Prior):
    r"""MAP prior of the group penalty gamma ||x||_{2,1}: group soft-threshold along ``axis``.

    tramp/priors/map_L21_norm_prior.py:22-80 with ``axis=0``: ``size = (d,) + site shape``, the groups are
    the d components of a site - the total-variation prior on a gradient field
    (tramp/models/total_variation_model.py:28-40).  Device vectors are [batch, d n].
    """

    def __init__(self, size, gamma=1, axis=0, isotropic=True, dtype=np.float64, ctx=None):
        assert type(size) == tuple and len(size) > 1, "size must be a tuple of length > 1"
        if axis != 0:
            raise NotImplementedError("MAP_L21NormPrior on the HIP path groups along axis 0")
        if not isotropic:
            raise NotImplementedError("diagonal beliefs for MAP_L21NormPrior are outside the HIP path")
        self.size = size
        self.gamma = gamma
        self.axis = axis
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self.N = int(np.prod(size))
        self.d = size[axis]
        self.n_sites = self.N // self.d

    def sample(self):
        import warnings
        warnings.warn("MAP_L21NormPrior.sample not implemented return zero array as a placeholder")
        return np.zeros(self.size)                              # map_L21_norm_prior.py:53-58

    def second_moment(self):
        raise NotImplementedError

    def _vec(self, bx):
        if isinstance(bx, DeviceArray):
            return VecArg(self.ctx, bx, self.dtype)
        h = np.asarray(bx, dtype=self.dtype)
        nd = len(self.size)
        if h.shape[-nd:] != tuple(self.size) or h.ndim not in (nd, nd + 1):
            raise ValueError(f"expected an array of shape {self.size} or (batch,)+{self.size}, got {h.shape}")
        arg = VecArg(self.ctx, h.reshape(-1, self.N) if h.ndim == nd + 1 else h.reshape(self.N), self.dtype)
        arg.host_shape = h.shape
        arg._one = h.ndim == nd
        return arg

    def _run(self, ax, bx, out_a=None, out_b=None, damping=0.0, old_a=None, old_b=None, want_posterior=False,
             want_support=False):
        ctx = self.ctx
        b = self._vec(bx)
        if b.n != self.N:
            raise ValueError(f"expected messages of {self.N} components, got {b.n}")
        a = ScalArg(ctx, ax, b.batch)
        a_new = out_a if out_a is not None else ctx.empty((b.batch,), np.float64)
        b_new = out_b if out_b is not None else ctx.empty((b.batch, self.N), self.dtype)
        r = ctx.empty((b.batch, self.N), self.dtype) if want_posterior else None
        v = ctx.empty((b.batch,), np.float64) if want_posterior else None
        count = ctx.empty((b.batch,), np.uint64) if want_support else None
        mask = ctx.empty((b.batch, self.n_sites), np.uint8) if want_support else None
        ctx._call("tramp_l21_prior_fwd", self._dt, self.d, self.n_sites, b.batch, float(self.gamma), a.dev.ptr,
                  b.dev.ptr, float(damping or 0.0), old_a.ptr if old_a is not None else None,
                  old_b.ptr if old_b is not None else None, a_new.ptr, b_new.ptr,
                  r.ptr if r is not None else None, v.ptr if v is not None else None,
                  count.ptr if count is not None else None, mask.ptr if mask is not None else None)
        return a, b, a_new, b_new, r, v, count, mask

    @staticmethod
    def _ret(dev, like):
        return dev.get().reshape(like.host_shape) if like.was_host else dev

    @staticmethod
    def _ret_scal(dev, a, like):
        if a.was_host and like.was_host:
            h = dev.get()
            return float(h[0]) if (a.scalar_like and getattr(like, "_one", False)) else h
        return dev

    def compute_forward_posterior(self, ax, bx):
        a, b, _, _, r, v, _, _ = self._run(ax, bx, want_posterior=True)          # :69-74
        return self._ret(r, b), self._ret_scal(v, a, b)

    def compute_forward_message(self, ax, bx, **kw):
        a, b, a_new, b_new, _, _, _, _ = self._run(ax, bx, **kw)
        return self._ret_scal(a_new, a, b), self._ret(b_new, b)

    def support(self, ax, bx):
        "sites with ||b[:, j]|| > gamma (v_group_soft_threshold, :17-20): exact mask and its count"
        a, b, _, _, _, _, count, mask = self._run(ax, bx, want_support=True)
        m = mask.get().astype(bool)
        site_shape = tuple(self.size[1:])
        return (m.reshape(site_shape) if getattr(b, "_one", False) else m.reshape((b.batch,) + site_shape)), count.get()

    def compute_forward_error(self, ax):
        raise NotImplementedError      # as the reference (map_L21_norm_prior.py:90-97)


class GaussianPrior(Prior):
    r"""Gaussian prior N(x | mean, var): constant message, closed-form errors.

    tramp/priors/gaussian_prior.py:8-143 (kernel K3).
    """

    def __init__(self, size, mean=0, var=1, isotropic=True, dtype=np.float64, ctx=None):
        self.size = size
        self.mean = mean
        self.var = var
        self.isotropic = isotropic
        self.repr_init()
        self._init_device(dtype, ctx)
        self.sigma = np.sqrt(var)
        self.a = 1 / var
        self.b = mean / var

    _diag_which = _lib.DIAG_GAUSS_PRIOR

    def _diag_params(self):
        return (self.mean, self.var)

    def sample(self):
        return self.mean + self.sigma * np.random.standard_normal(self.size)

    def second_moment(self):
        return self.mean**2 + self.var

    def compute_log_partition(self, ax, bx):
        "gaussian_prior.py:70-74"
        return self._logz(_lib.LOGZ_GAUSS_PRIOR, ax, bx, params=(self.mean, self.var))

    def compute_forward_posterior(self, ax, bx):
        """gaussian_prior.py:63-68: r = (bx + mean/var) / (ax + 1/var), v = 1 / (ax + 1/var) - the product of
        two Gaussian messages, i.e. kernel K11 on the incoming message and the prior's own constant one."""
        if not self.isotropic:
            _, _, r, v = self._diag(ax, bx)
            return r, v
        ctx = self.ctx
        b = VecArg(ctx, bx, self.dtype)
        a = ScalArg(ctx, ax, b.batch)
        a0, b0 = ctx.empty((b.batch,), np.float64), out_vec(ctx, b, self.dtype)
        ctx._call("tramp_gauss_prior_fwd", self._dt, b.n, b.batch, float(self.mean), float(self.var), a0, b0)
        r, v = out_vec(ctx, b, self.dtype), ctx.empty((b.batch,), np.float64)
        ctx._call("tramp_variable_posterior", self._dt, b.n, b.batch, a.dev, b.dev, a0, b0, None, r, v, None, None)
        return ret_vec(r, b), ret_scal(v, a, b)

    def compute_forward_message(self, ax, bx, out_a=None, out_b=None, **_):
        "gaussian_prior.py:86-89: bypasses compute_ab_new (and damping sees a constant)"
        if not self.isotropic:
            return self.a * np.ones_like(ax), self.b * np.ones_like(bx)
        ctx = self.ctx
        b = VecArg(ctx, bx, self.dtype)
        a = ScalArg(ctx, ax, b.batch)
        a_new = out_a if out_a is not None else ctx.empty((b.batch,), np.float64)
        b_new = out_b if out_b is not None else out_vec(ctx, b, self.dtype)
        ctx._call("tramp_gauss_prior_fwd", self._dt, b.n, b.batch, float(self.mean), float(self.var),
                  a_new.ptr, b_new.ptr)
        return ret_scal(a_new, a, b), ret_vec(b_new, b)

    def compute_forward_error(self, ax):
        "gaussian_prior.py:76-79"
        return 1 / (ax + self.a)

    def compute_forward_state_evolution(self, ax):
        "gaussian_prior.py:91-93"
        return self.a


PRIOR_CLASSES = {
    "gauss_bernoulli": GaussBernoulliPrior,
    "gaussian": GaussianPrior,
    "L1_norm": MAP_L1NormPrior,
    "L21_norm": MAP_L21NormPrior,
    "binary": BinaryPrior,
    "exponential": ExponentialPrior,
    "positive": ExponentialPrior,          # as found in the reference's registry (tramp/priors/__init__.py:19)
    "mixture": GaussianMixturePrior,
}


def get_prior(size, prior_type, **kwargs):
    "tramp/priors/__init__.py:25-27"
    return PRIOR_CLASSES[prior_type](size=size, **kwargs)
