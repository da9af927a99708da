"""Bind the REFERENCE's own factor classes to libtramp_hip.so, so that `tramp.algos.ExpectationPropagation(model).iterate()`
(tramp/algos/expectation_propagation.py:11-15 -> Factor.forward_message / backward_message, tramp/base.py:329-365) runs
its per-factor arithmetic through the C ABI of include/tramp_hip.h - the executable form of INTEGRATION.md section 2.

    import tramp                      # the reference, unmodified
    from tramp_amd import compat
    binding = compat.bind_reference() # overrides compute_*_message on the reference's classes
    ExpectationPropagation(model).iterate(max_iter=30)          # tramp.algos: every factor update is now a HIP kernel
    binding.unbind()

The reference's scheduler, message dictionaries, damping, callbacks and variables stay as they are (host, NumPy): each
overridden method uploads its arguments, makes ONE C-ABI call and downloads (a_new, b_new) - "strict mode" of SURVEY 8b.
It is the drop-in proof and the migration path, not the fast path: `tramp_amd.ExpectationPropagation` keeps the messages
on the device.  tramp's model layer `isinstance`-checks its nodes (`tramp/models/dag_algebra.py:215-219`), which is why
the binding patches the reference's classes instead of substituting `tramp_amd`'s.

What is bound (reference method -> entry point):
    GaussBernoulliPrior.compute_forward_message   (priors/base_prior.py:13-16, gauss_bernoulli_prior.py:70-77) -> tramp_gb_prior_fwd
    GaussianPrior.compute_forward_message         (priors/gaussian_prior.py:86-89)                            -> tramp_gauss_prior_fwd
    MAP_L1NormPrior.compute_forward_message       (priors/map_L1_norm_prior.py:66-71)                         -> tramp_l1_prior_fwd
    GaussianLikelihood.compute_backward_message   (likelihoods/gaussian_likelihood.py:68-71)                  -> tramp_gauss_lik_bwd
    SgnLikelihood / AbsLikelihood.compute_backward_message (likelihoods/base_likelihood.py:25-28)             -> tramp_sgn_lik_bwd / tramp_abs_lik_bwd
    SgnChannel / AbsChannel / ReluChannel.compute_forward_message, compute_backward_message
                                                  (channels/base_channel.py:9-17, piecewise_linear_channel.py:27-53) -> tramp_pwl_channel
    LinearChannel.__init__ (keeps the reference's own SVD, uploads its thin factors once)                     -> tramp_linear_create
    LinearChannel.compute_forward_message, compute_backward_message (linear_channel.py:69-117)                -> tramp_linear_message
  round 6:
    BinaryPrior.compute_forward_message           (priors/binary_prior.py:52-58)                              -> tramp_binary_prior_fwd
    ExponentialPrior / PositivePrior.compute_forward_message (priors/exponential_prior.py:54-60, positive_prior.py:56-63)
                                                                                                               -> tramp_halfline_prior_fwd
    GaussianMixturePrior.compute_forward_message  (priors/gaussian_mixture_prior.py:88-97)                    -> tramp_gm_prior_fwd
    ModulusLikelihood.compute_backward_message    (likelihoods/modulus_likelihood.py:82-93)                   -> tramp_modulus_lik_bwd
    ComplexLinearChannel.__init__ / compute_*_message (channels/linear/complex_linear_channel.py:36-141): the thin factors of
        the reference's complex SVD in their real embedding [[Re, -Im], [Im, Re]]              -> tramp_linear_create / _message

`Backend(lib=...)` takes any object with the library's entry points as attributes: the real `ctypes.CDLL` (default) or
the recording stand-in of tests/test_compat_reference.py, which checks every call against `tramp_amd._lib._SIGNATURES`
where the reference lives and no GPU does.
"""
import ctypes as C

import numpy as np

from . import _lib as L

vp, sz = C.c_void_p, C.c_size_t
KIND_OF = {"SgnChannel": L.SGN, "AbsChannel": L.ABS, "ReluChannel": L.RELU}


class Backend:
    """One context of the C ABI and the strict-mode calls (host arrays in, host arrays out).  Every entry point is called
    positionally, with exactly the argument list of include/tramp_hip.h."""

    def __init__(self, lib=None, device=0):
        self.lib = lib if lib is not None else L.load()
        self.ctx = vp()
        self._check(self.lib.tramp_ctx_create(int(device), C.byref(self.ctx)), "tramp_ctx_create")
        self._live = []

    # ---- plumbing: tramp_buf_alloc / tramp_h2d / tramp_d2h / tramp_buf_free
    def _check(self, rc, where):
        if rc != L.OK:
            detail = ""
            if self.ctx:
                msg = self.lib.tramp_last_error(self.ctx)
                detail = msg.decode(errors="replace") if isinstance(msg, bytes) else str(msg or "")
            raise L.TrampHipError(rc, where, detail)

    def alloc(self, nbytes):
        p = vp()
        self._check(self.lib.tramp_buf_alloc(self.ctx, sz(max(int(nbytes), 8)), C.byref(p)), "tramp_buf_alloc")
        return p

    def free(self, p):
        self._check(self.lib.tramp_buf_free(self.ctx, p), "tramp_buf_free")

    def upload(self, host, dtype=np.float64):
        host = np.ascontiguousarray(host, dtype=dtype)
        p = self.alloc(host.nbytes)
        if host.nbytes:
            self._check(self.lib.tramp_h2d(self.ctx, p, vp(host.ctypes.data), sz(host.nbytes)), "tramp_h2d")
        return p

    def download(self, p, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            self._check(self.lib.tramp_d2h(self.ctx, vp(out.ctypes.data), p, sz(out.nbytes)), "tramp_d2h")
        return out

    class _Scope:
        "device buffers of one call, freed on exit"

        def __init__(self, be):
            self.be, self.ptrs = be, []

        def up(self, host):
            p = self.be.upload(host)
            self.ptrs.append(p)
            return p

        def out(self, count):
            p = self.be.alloc(8 * int(count))
            self.ptrs.append(p)
            return p

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            for p in self.ptrs:
                self.be.free(p)
            return False

    def _finish(self, s, d_a, d_b, n, shape):
        a_new = float(self.download(d_a, (1,))[0])
        b_new = self.download(d_b, (n,)).reshape(shape)
        return a_new, b_new

    # ---- priors: (ax, bx) -> (ax_new, bx_new)
    def gb_prior_forward_message(self, ax, bx, rho, mean, var):
        bx = np.asarray(bx, dtype=np.float64)
        n = bx.size
        with self._Scope(self) as s:
            d_ax, d_bx, d_a, d_b = s.up([ax]), s.up(bx.ravel()), s.out(1), s.out(n)
            self._check(self.lib.tramp_gb_prior_fwd(self.ctx, L.F64, sz(n), sz(1), d_ax, d_bx, float(rho), float(mean), float(var),
                                                    0.0, None, None, d_a, d_b, None, None), "tramp_gb_prior_fwd")
            return self._finish(s, d_a, d_b, n, bx.shape)

    def l1_prior_forward_message(self, ax, bx, gamma):
        bx = np.asarray(bx, dtype=np.float64)
        n = bx.size
        with self._Scope(self) as s:
            d_ax, d_bx, d_a, d_b = s.up([ax]), s.up(bx.ravel()), s.out(1), s.out(n)
            self._check(self.lib.tramp_l1_prior_fwd(self.ctx, L.F64, sz(n), sz(1), d_ax, d_bx, float(gamma), 0.0, None, None,
                                                    d_a, d_b, None, None, None, None), "tramp_l1_prior_fwd")
            return self._finish(s, d_a, d_b, n, bx.shape)

    def gauss_prior_forward_message(self, bx, mean, var):
        bx = np.asarray(bx, dtype=np.float64)
        n = bx.size
        with self._Scope(self) as s:
            d_a, d_b = s.out(1), s.out(n)
            self._check(self.lib.tramp_gauss_prior_fwd(self.ctx, L.F64, sz(n), sz(1), float(mean), float(var), d_a, d_b),
                        "tramp_gauss_prior_fwd")
            return self._finish(s, d_a, d_b, n, bx.shape)

    def binary_prior_forward_message(self, ax, bx, p_pos):
        bx = np.asarray(bx, dtype=np.float64)
        n = bx.size
        with self._Scope(self) as s:
            d_ax, d_bx, d_a, d_b = s.up([ax]), s.up(bx.ravel()), s.out(1), s.out(n)
            self._check(self.lib.tramp_binary_prior_fwd(self.ctx, L.F64, sz(n), sz(1), d_ax, d_bx, float(0.5 * np.log(p_pos / (1 - p_pos))),
                                                        0.0, None, None, d_a, d_b, None, None), "tramp_binary_prior_fwd")
            return self._finish(s, d_a, d_b, n, bx.shape)

    def halfline_prior_forward_message(self, ax, bx, a0, b0):
        "ExponentialPrior (a0 = 0, b0 = -1/mean) and PositivePrior (a0 = 1, b0 = 0)"
        bx = np.asarray(bx, dtype=np.float64)
        n = bx.size
        with self._Scope(self) as s:
            d_ax, d_bx, d_a, d_b = s.up([ax]), s.up(bx.ravel()), s.out(1), s.out(n)
            self._check(self.lib.tramp_halfline_prior_fwd(self.ctx, L.F64, sz(n), sz(1), d_ax, d_bx, float(a0), float(b0), 0.0, None,
                                                          None, d_a, d_b, None, None), "tramp_halfline_prior_fwd")
            return self._finish(s, d_a, d_b, n, bx.shape)

    def gm_prior_forward_message(self, ax, bx, probs, means, vars_):
        bx = np.asarray(bx, dtype=np.float64)
        n, K = bx.size, len(probs)
        tabs = [(C.c_double * K)(*[float(x) for x in t]) for t in (probs, means, vars_)]
        with self._Scope(self) as s:
            d_ax, d_bx, d_a, d_b = s.up([ax]), s.up(bx.ravel()), s.out(1), s.out(n)
            self._check(self.lib.tramp_gm_prior_fwd(self.ctx, L.F64, sz(n), sz(1), d_ax, d_bx, int(K), tabs[0], tabs[1], tabs[2], 0.0,
                                                    None, None, d_a, d_b, None, None), "tramp_gm_prior_fwd")
            return self._finish(s, d_a, d_b, n, bx.shape)

    # ---- likelihoods: (az, bz; y) -> (az_new, bz_new)
    def modulus_lik_backward_message(self, az, bz, y):
        "bz planar (2, n) (tramp/utils/misc.py:13-27), y = |z| of n components"
        bz, y = np.asarray(bz, dtype=np.float64), np.asarray(y, dtype=np.float64)
        n = y.size
        assert bz.shape == (2,) + y.shape, "the message of a ModulusLikelihood is the planar array (2,) + y.shape"
        with self._Scope(self) as s:
            d_az, d_bz, d_y, d_a, d_b = s.up([az]), s.up(bz.ravel()), s.up(y.ravel()), s.out(1), s.out(2 * n)
            self._check(self.lib.tramp_modulus_lik_bwd(self.ctx, L.F64, sz(n), sz(1), d_az, d_bz, d_y, 0.0, None, None, d_a, d_b,
                                                       None, None), "tramp_modulus_lik_bwd")
            return self._finish(s, d_a, d_b, 2 * n, bz.shape)

    def gauss_lik_backward_message(self, bz, y, var):
        y = np.asarray(y, dtype=np.float64)
        n = y.size
        with self._Scope(self) as s:
            d_y, d_a, d_b = s.up(y.ravel()), s.out(1), s.out(n)
            self._check(self.lib.tramp_gauss_lik_bwd(self.ctx, L.F64, sz(n), sz(1), d_y, float(var), d_a, d_b),
                        "tramp_gauss_lik_bwd")
            return self._finish(s, d_a, d_b, n, y.shape)

    def _lik(self, symbol, az, bz, y):
        bz = np.asarray(bz, dtype=np.float64)
        n = bz.size
        with self._Scope(self) as s:
            d_az, d_bz, d_y, d_a, d_b = s.up([az]), s.up(bz.ravel()), s.up(np.asarray(y, dtype=np.float64).ravel()), s.out(1), s.out(n)
            self._check(getattr(self.lib, symbol)(self.ctx, L.F64, sz(n), sz(1), d_az, d_bz, d_y, 0.0, None, None, d_a, d_b,
                                                  None, None), symbol)
            return self._finish(s, d_a, d_b, n, bz.shape)

    def sgn_lik_backward_message(self, az, bz, y):
        return self._lik("tramp_sgn_lik_bwd", az, bz, y)

    def abs_lik_backward_message(self, az, bz, y):
        return self._lik("tramp_abs_lik_bwd", az, bz, y)

    # ---- componentwise channels: (az, bz, ax, bx) -> the x-side (FWD) or z-side (BWD) message
    def pwl_channel_message(self, kind, direction, az, bz, ax, bx):
        bz, bx = np.asarray(bz, dtype=np.float64), np.asarray(bx, dtype=np.float64)
        n = bz.size
        with self._Scope(self) as s:
            d_az, d_bz, d_ax, d_bx, d_a, d_b = s.up([az]), s.up(bz.ravel()), s.up([ax]), s.up(bx.ravel()), s.out(1), s.out(n)
            self._check(self.lib.tramp_pwl_channel(self.ctx, L.F64, int(kind), int(direction), sz(n), sz(1), d_az, d_bz, d_ax, d_bx,
                                                   0.0, None, None, d_a, d_b, None, None, None), "tramp_pwl_channel")
            return self._finish(s, d_a, d_b, n, (bx if direction == L.FWD else bz).shape)

    # ---- dense linear channel
    def linear_create(self, U, S, V, rank):
        """thin factors of the reference's own SVD W = U S V^T (linear_channel.py:8-15): Uk [M, k], Vt [k, N], sing [k],
        uploaded once; the handle keeps the device pointers (include/tramp_hip.h: tramp_linear_create)"""
        M, N = U.shape[0], V.shape[0]
        k = min(M, N)
        sing = np.zeros(k)
        d = np.diag(S)
        sing[:d.size] = d[:k]
        bufs = [self.upload(np.ascontiguousarray(U[:, :k])), self.upload(np.ascontiguousarray(V[:, :k].T)), self.upload(sing)]
        h = vp()
        self._check(self.lib.tramp_linear_create(self.ctx, L.F64, sz(M), sz(N), sz(int(rank)), bufs[0], bufs[1], bufs[2], C.byref(h)),
                    "tramp_linear_create")
        return h, bufs, (M, N)

    def complex_linear_create(self, U, S, V, rank):
        """ComplexLinearChannel: the reference's complex SVD W = U S V^H (complex_linear_channel.py:8-15) in the real
        embedding E(A) = [[Re A, -Im A], [Im A, Re A]] (E is multiplicative and E(A^H) = E(A)^T): W_r = E(U_k) diag(s, s)
        E(V_k)^T acts on planar vectors [Re; Im] - the same thin-factor handle as a real channel of shape [2M, 2N]"""
        M, N = U.shape[0], V.shape[0]
        k = min(M, N)
        sing = np.zeros(k)
        d = np.real(np.diag(S))
        sing[:d.size] = d[:k]

        def emb(A):
            return np.block([[A.real, -A.imag], [A.imag, A.real]])
        Uk, Vk = emb(U[:, :k]), emb(V[:, :k])
        bufs = [self.upload(np.ascontiguousarray(Uk)), self.upload(np.ascontiguousarray(Vk.T)), self.upload(np.concatenate([sing, sing]))]
        h = vp()
        self._check(self.lib.tramp_linear_create(self.ctx, L.F64, sz(2 * M), sz(2 * N), sz(2 * int(rank)), bufs[0], bufs[1], bufs[2],
                                                 C.byref(h)), "tramp_linear_create")
        return h, bufs, (2 * M, 2 * N)

    def linear_destroy(self, handle):
        h, bufs, _ = handle
        self._check(self.lib.tramp_linear_destroy(self.ctx, h), "tramp_linear_destroy")
        for p in bufs:
            self.free(p)

    def linear_message(self, handle, direction, az, bz, ax, bx):
        h, _, (M, N) = handle
        bz, bx = np.asarray(bz, dtype=np.float64), np.asarray(bx, dtype=np.float64)
        assert bz.size == N and bx.size == M, "message sizes do not match the channel"
        n = M if direction == L.FWD else N
        with self._Scope(self) as s:
            d_az, d_bz, d_ax, d_bx, d_a, d_b = s.up([az]), s.up(bz.ravel()), s.up([ax]), s.up(bx.ravel()), s.out(1), s.out(n)
            self._check(self.lib.tramp_linear_message(self.ctx, h, int(direction), sz(1), d_az, d_bz, d_ax, d_bx, 0.0, None, None,
                                                      d_a, d_b, None, None, 0), "tramp_linear_message")
            return self._finish(s, d_a, d_b, n, (bx if direction == L.FWD else bz).shape)

    def close(self):
        if self.ctx:
            self.lib.tramp_ctx_destroy(self.ctx)
            self.ctx = vp()


class Binding:
    "what bind_reference() installed; unbind() puts the reference's own methods back"

    def __init__(self, backend):
        self.backend = backend
        self._saved = []

    def _patch(self, cls, name, fn):
        self._saved.append((cls, name, cls.__dict__.get(name)))
        setattr(cls, name, fn)

    def unbind(self):
        for cls, name, old in reversed(self._saved):
            if old is None:
                delattr(cls, name)
            else:
                setattr(cls, name, old)
        self._saved = []


def bind_reference(backend=None):
    """Install the overrides on the classes of the importable `tramp` package.  Raises ImportError if the reference is
    not on sys.path.  Returns the Binding (unbind() restores)."""
    import tramp.channels as tc
    import tramp.likelihoods as tl
    import tramp.priors as tp
    be = backend if backend is not None else Backend()
    b = Binding(be)

    def gb_fwd(self, ax, bx):
        return be.gb_prior_forward_message(ax, bx, self.rho, self.mean, self.var)

    def gauss_prior_fwd(self, ax, bx):
        return be.gauss_prior_forward_message(bx, self.mean, self.var)

    def l1_fwd(self, ax, bx):
        return be.l1_prior_forward_message(ax, bx, self.gamma)

    def gauss_lik_bwd(self, az, bz):
        return be.gauss_lik_backward_message(bz, self.y, self.var)

    def sgn_lik_bwd(self, az, bz):
        return be.sgn_lik_backward_message(az, bz, self.y)

    def abs_lik_bwd(self, az, bz):
        return be.abs_lik_backward_message(az, bz, self.y)

    b._patch(tp.GaussBernoulliPrior, "compute_forward_message", gb_fwd)
    b._patch(tp.GaussianPrior, "compute_forward_message", gauss_prior_fwd)
    b._patch(tp.MAP_L1NormPrior, "compute_forward_message", l1_fwd)
    b._patch(tl.GaussianLikelihood, "compute_backward_message", gauss_lik_bwd)
    b._patch(tl.SgnLikelihood, "compute_backward_message", sgn_lik_bwd)
    b._patch(tl.AbsLikelihood, "compute_backward_message", abs_lik_bwd)

    for cls in (tc.SgnChannel, tc.AbsChannel, tc.ReluChannel):
        kind = KIND_OF[cls.__name__]

        def fwd(self, az, bz, ax, bx, _k=kind):
            return be.pwl_channel_message(_k, L.FWD, az, bz, ax, bx)

        def bwd(self, az, bz, ax, bx, _k=kind):
            return be.pwl_channel_message(_k, L.BWD, az, bz, ax, bx)

        b._patch(cls, "compute_forward_message", fwd)
        b._patch(cls, "compute_backward_message", bwd)

    lin_init = tc.LinearChannel.__init__

    def linear_init(self, W, precompute_svd=True, name="W"):
        lin_init(self, W, precompute_svd=precompute_svd, name=name)
        self._hip = be.linear_create(self.U, self.S, self.V, self.rank) if precompute_svd else None

    def _handle(self):
        h = getattr(self, "_hip", None)
        if h is None and getattr(self, "precompute_svd", False):          # built before bind_reference(): bind lazily
            h = self._hip = be.linear_create(self.U, self.S, self.V, self.rank)
        return h

    lin_fwd, lin_bwd = tc.LinearChannel.compute_forward_message, tc.LinearChannel.compute_backward_message

    def linear_fwd(self, az, bz, ax, bx):
        h = _handle(self)
        return lin_fwd(self, az, bz, ax, bx) if h is None else be.linear_message(h, L.FWD, az, bz, ax, bx)

    def linear_bwd(self, az, bz, ax, bx):
        h = _handle(self)
        return lin_bwd(self, az, bz, ax, bx) if h is None else be.linear_message(h, L.BWD, az, bz, ax, bx)

    b._patch(tc.LinearChannel, "__init__", linear_init)
    b._patch(tc.LinearChannel, "compute_forward_message", linear_fwd)
    b._patch(tc.LinearChannel, "compute_backward_message", linear_bwd)

    # ---- round 6: the rest of the separable priors and the complex factors of phase retrieval
    def binary_fwd(self, ax, bx):
        return be.binary_prior_forward_message(ax, bx, self.p_pos)

    def exp_fwd(self, ax, bx):
        return be.halfline_prior_forward_message(ax, bx, 0.0, self.b)           # exponential_prior.py:26, 54-60

    def positive_fwd(self, ax, bx):
        return be.halfline_prior_forward_message(ax, bx, self.a, self.b)         # positive_prior.py:23-25, 56-63

    def gm_fwd(self, ax, bx):
        return be.gm_prior_forward_message(ax, bx, self.probs, self.means, self.vars)

    def modulus_bwd(self, az, bz):
        return be.modulus_lik_backward_message(az, bz, self.y)

    b._patch(tp.BinaryPrior, "compute_forward_message", binary_fwd)
    b._patch(tp.ExponentialPrior, "compute_forward_message", exp_fwd)
    b._patch(tp.PositivePrior, "compute_forward_message", positive_fwd)
    b._patch(tp.GaussianMixturePrior, "compute_forward_message", gm_fwd)
    b._patch(tl.ModulusLikelihood, "compute_backward_message", modulus_bwd)

    clin_init = tc.ComplexLinearChannel.__init__
    clin_fwd, clin_bwd = tc.ComplexLinearChannel.compute_forward_message, tc.ComplexLinearChannel.compute_backward_message

    def complex_init(self, W, precompute_svd=True, name="W"):
        clin_init(self, W, precompute_svd=precompute_svd, name=name)
        self._hip = be.complex_linear_create(self.U, self.S, self.V, self.rank) if precompute_svd else None

    def _chandle(self):
        h = getattr(self, "_hip", None)
        if h is None and getattr(self, "precompute_svd", False):
            h = self._hip = be.complex_linear_create(self.U, self.S, self.V, self.rank)
        return h

    def _planar_message(h, direction, az, bz, ax, bx):
        bz, bx = np.asarray(bz, dtype=np.float64), np.asarray(bx, dtype=np.float64)
        a_new, b_new = be.linear_message(h, direction, az, bz.reshape(-1), ax, bx.reshape(-1))
        return a_new, b_new.reshape((bx if direction == L.FWD else bz).shape)

    def complex_fwd(self, az, bz, ax, bx):
        h = _chandle(self)
        return clin_fwd(self, az, bz, ax, bx) if h is None else _planar_message(h, L.FWD, az, bz, ax, bx)

    def complex_bwd(self, az, bz, ax, bx):
        h = _chandle(self)
        return clin_bwd(self, az, bz, ax, bx) if h is None else _planar_message(h, L.BWD, az, bz, ax, bx)

    b._patch(tc.ComplexLinearChannel, "__init__", complex_init)
    b._patch(tc.ComplexLinearChannel, "compute_forward_message", complex_fwd)
    b._patch(tc.ComplexLinearChannel, "compute_backward_message", complex_bwd)
    return b


# ---------------------------------------------------------------------------------------------------------------------
# The other direction: a script written against the reference's package layout runs on this package.
_ALIASES = ("algos", "base", "channels", "ensembles", "experiments", "likelihoods", "models", "priors", "variables")
_SUBMODULE_ALIASES = {"algos.metrics": "metrics", "algos.callbacks": "algos", "algos.initial_conditions": "algos",
                      "priors.base_prior": "priors", "likelihoods.base_likelihood": "likelihoods",
                      "channels.base_channel": "channels", "models.base_model": "models"}


def install_as_tramp(name="tramp"):
    """Register this package under the reference's import paths: after ``tramp_amd.compat.install_as_tramp()`` the lines
    ``from tramp.models import glm_generative``, ``from tramp.experiments import BayesOptimalScenario``,
    ``from tramp.algos.metrics import mean_squared_error`` ... of a script written for the reference import THIS package
    (every name examples/ of the reference imports, except the matplotlib helpers of tramp/experiments/plots.py).  Refuses when a
    different ``tramp`` is already imported - use bind_reference() to keep the reference's scheduler on the HIP kernels instead."""
    import importlib
    import sys
    pkg = importlib.import_module(__package__)
    have = sys.modules.get(name)
    if have is not None and have is not pkg:
        raise ImportError(f"a different {name!r} package is already imported from {getattr(have, '__file__', '?')}")
    sys.modules[name] = pkg
    for sub in _ALIASES:
        sys.modules[f"{name}.{sub}"] = importlib.import_module(f"{__package__}.{sub}")
    for path, target in _SUBMODULE_ALIASES.items():
        mod = importlib.import_module(f"{__package__}.{target}")
        sys.modules[f"{name}.{path}"] = mod
        sys.modules[f"{__package__}.{path}"] = mod           # tramp_amd.algos.metrics ... resolve as well
        parent, leaf = path.rsplit(".", 1)
        setattr(importlib.import_module(f"{__package__}.{parent}"), leaf, mod)
    return pkg
