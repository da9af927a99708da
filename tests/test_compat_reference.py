"""The drop-in claim against the REFERENCE'S OWN scheduler (build container only: needs /root/reference; no GPU here).

`tramp_amd.compat.bind_reference()` overrides compute_*_message on the reference's classes so that
`tramp.algos.ExpectationPropagation(model).iterate()` (tramp/algos/expectation_propagation.py:11-15,
tramp/base.py:329-365) makes one C-ABI call per factor update.  Here the library is a RECORDING STAND-IN: an object with
one attribute per symbol of `tramp_amd._lib._SIGNATURES` that
  * refuses a call whose argument count differs from the table, or whose arguments ctypes would not convert to the
    table's argtypes (the conversion the real CDLL performs),
  * tracks every "device" allocation and checks the size of every buffer an entry point reads or writes
    (n x batch x 8 bytes for messages, batch x 8 for precisions) against what was allocated,
  * computes the result with the oracle (NumPy restatement) and writes it into the caller's output buffers.
Then the reference's unmodified `iterate()` runs the sparse GLM, the 1-bit CS model and the two-layer ReLU model of
tools/make_golden.py twice - on its own NumPy methods and through the binding - and must (a) produce the same messages
and estimates and (b) have issued exactly the expected sequence of entry points.  On a GPU box the same Backend runs
against the real library (tests/test_gpu_compat.py).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import tramp_oracle as orc

REF = os.environ.get("TRAMP_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tramp")), reason="the reference is not on this machine")


@pytest.fixture(scope="module")
def ref():
    "the reference package, with the networkx-1.x shim of tools/make_golden.py (SURVEY 8c)"
    import networkx as nx
    if not getattr(nx.DiGraph, "_tramp_shim", False):
        class DiGraph1(nx.DiGraph):
            _tramp_shim = True

            @property
            def node(self):
                return self._node

            def predecessors(self, n):
                return list(super().predecessors(n))

            def successors(self, n):
                return list(super().successors(n))

            def in_edges(self, *a, **k):
                return list(super().in_edges(*a, **k))

            def edges(self, *a, **k):
                return list(super().edges(*a, **k))

            def nodes(self, *a, **k):
                return list(super().nodes(*a, **k))

        _topo = nx.topological_sort
        nx.DiGraph = DiGraph1
        nx.topological_sort = lambda g: list(_topo(g))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import tramp.algos
    import tramp.channels
    import tramp.likelihoods
    import tramp.priors
    import tramp.variables
    return tramp


# ------------------------------------------------------------------------------------------------ the stand-in library
class FakeLib:
    """Recording stand-in for libtramp_hip.so.  "Device memory" is host memory: pointers are addresses of ctypes buffers."""

    def __init__(self):
        from tramp_amd import _lib as L
        self.L = L
        self.calls = []                    # (symbol, detail)
        self.buffers = {}                  # address -> (ctypes buffer, nbytes)
        self.handles = {}                  # id -> dict of a linear channel
        self.next_handle = 0x1000
        for name in L._SIGNATURES:
            if not hasattr(type(self), "_" + name):
                setattr(self, name, self._unbound(name))
            else:
                setattr(self, name, self._entry(name))

    # -- the checks the real CDLL + the device would make
    def _convert(self, name, args):
        sig = self.L._SIGNATURES[name]
        assert len(args) == len(sig), f"{name}: {len(args)} arguments, the header declares {len(sig)}"
        for i, (a, t) in enumerate(zip(args, sig)):
            try:
                t.from_param(a)
            except (TypeError, C.ArgumentError) as e:
                raise AssertionError(f"{name}: argument {i} ({a!r}) does not convert to {t.__name__}: {e}")

    def _entry(self, name):
        impl = getattr(self, "_" + name)

        def call(*args):
            self._convert(name, args)
            rc = impl(*args)
            self.calls.append(name)
            return 0 if rc is None else rc
        return call

    def _unbound(self, name):
        def call(*args):
            raise AssertionError(f"{name} was called, but the binding under test should not need it")
        return call

    @staticmethod
    def _addr(p):
        if p is None:
            return None
        if isinstance(p, int):
            return p
        return p.value

    def view(self, p, count, what):
        "numpy view of `count` doubles of a device buffer; the buffer must exist and be large enough"
        a = self._addr(p)
        assert a in self.buffers, f"{what}: {a} is not a live device buffer"
        buf, nbytes = self.buffers[a]
        assert 8 * count <= nbytes, f"{what}: entry point touches {8 * count} bytes of a {nbytes}-byte buffer"
        return np.frombuffer(buf, dtype=np.float64, count=count)

    # -- context and memory
    def _tramp_ctx_create(self, device, out):
        out._obj.value = 0xC0FFEE

    def _tramp_ctx_destroy(self, ctx):
        pass

    def tramp_last_error(self, ctx):
        return b""

    def _tramp_buf_alloc(self, ctx, nbytes, out):
        n = nbytes.value if hasattr(nbytes, "value") else int(nbytes)
        buf = (C.c_char * max(n, 8))()
        a = C.addressof(buf)
        self.buffers[a] = (buf, n)
        out._obj.value = a

    def _tramp_buf_free(self, ctx, p):
        a = self._addr(p)
        assert a in self.buffers, "tramp_buf_free of a pointer that is not live"
        del self.buffers[a]

    def _tramp_h2d(self, ctx, dst, src, nbytes):
        n = nbytes.value
        a = self._addr(dst)
        assert a in self.buffers and n <= self.buffers[a][1], "tramp_h2d past the end of the device buffer"
        C.memmove(a, self._addr(src), n)

    def _tramp_d2h(self, ctx, dst, src, nbytes):
        n = nbytes.value
        a = self._addr(src)
        assert a in self.buffers and n <= self.buffers[a][1], "tramp_d2h past the end of the device buffer"
        C.memmove(self._addr(dst), a, n)

    # -- factor updates, computed by the oracle
    @staticmethod
    def _n(x):
        return x.value if hasattr(x, "value") else int(x)

    def _out(self, a_new_d, b_new_d, n, a_new, b_new):
        self.view(a_new_d, 1, "a_new")[0] = a_new
        self.view(b_new_d, n, "b_new")[:] = b_new

    def _tramp_gb_prior_fwd(self, ctx, dtype, n, batch, ax_d, bx_d, rho, mean, var, damping, a_old, b_old, a_new_d, b_new_d, r_d, v_d):
        n, batch = self._n(n), self._n(batch)
        assert dtype == self.L.F64 and batch == 1 and damping == 0.0 and a_old is None and b_old is None and r_d is None and v_d is None
        ax, bx = self.view(ax_d, 1, "ax")[0], self.view(bx_d, n, "bx")
        self._out(a_new_d, b_new_d, n, *orc.gb_prior_message(ax, bx.copy(), rho=rho, mean=mean, var=var))

    def _tramp_l1_prior_fwd(self, ctx, dtype, n, batch, ax_d, bx_d, gamma, damping, a_old, b_old, a_new_d, b_new_d, r_d, v_d, cnt, mask):
        n = self._n(n)
        ax, bx = self.view(ax_d, 1, "ax")[0], self.view(bx_d, n, "bx")
        self._out(a_new_d, b_new_d, n, *orc.l1_prior_message(ax, bx.copy(), gamma=gamma))

    def _tramp_gauss_prior_fwd(self, ctx, dtype, n, batch, mean, var, a_new_d, b_new_d):
        n = self._n(n)
        self._out(a_new_d, b_new_d, n, *orc.gauss_prior_message(0.0, np.zeros(n), mean=mean, var=var))

    def _tramp_gauss_lik_bwd(self, ctx, dtype, n, batch, y_d, var, a_new_d, b_new_d):
        n = self._n(n)
        y = self.view(y_d, n, "y")
        self._out(a_new_d, b_new_d, n, *orc.gauss_lik_message(0.0, np.zeros(n), y.copy(), var=var))

    def _lik(self, fn, n, az_d, bz_d, y_d, a_new_d, b_new_d):
        n = self._n(n)
        az, bz, y = self.view(az_d, 1, "az")[0], self.view(bz_d, n, "bz"), self.view(y_d, n, "y")
        self._out(a_new_d, b_new_d, n, *fn(az, bz.copy(), y.copy()))

    def _tramp_sgn_lik_bwd(self, ctx, dtype, n, batch, az_d, bz_d, y_d, damping, a_old, b_old, a_new_d, b_new_d, r_d, v_d):
        self._lik(orc.sgn_lik_message, n, az_d, bz_d, y_d, a_new_d, b_new_d)

    def _tramp_abs_lik_bwd(self, ctx, dtype, n, batch, az_d, bz_d, y_d, damping, a_old, b_old, a_new_d, b_new_d, r_d, v_d):
        self._lik(orc.abs_lik_message, n, az_d, bz_d, y_d, a_new_d, b_new_d)

    # round 6
    def _tramp_binary_prior_fwd(self, ctx, dtype, n, batch, ax_d, bx_d, b0, damping, a_old, b_old, a_new_d, b_new_d, r_d, v_d):
        n = self._n(n)
        ax, bx = self.view(ax_d, 1, "ax")[0], self.view(bx_d, n, "bx")
        r = np.tanh(bx + b0)                                   # binary_prior.py:52-58 with the natural parameter itself
        self._out(a_new_d, b_new_d, n, *orc.ab_new(r, (1 - r**2).mean(), ax, bx.copy()))

    def _tramp_halfline_prior_fwd(self, ctx, dtype, n, batch, ax_d, bx_d, a0, b0, damping, a_old, b_old, a_new_d, b_new_d, r_d, v_d):
        n = self._n(n)
        ax, bx = self.view(ax_d, 1, "ax")[0], self.view(bx_d, n, "bx").copy()
        r, v = orc.positive_rv(ax + a0, bx + b0)
        self._out(a_new_d, b_new_d, n, *orc.ab_new(r, v.mean(), ax, bx))

    def _tramp_gm_prior_fwd(self, ctx, dtype, n, batch, ax_d, bx_d, K, probs, means, vars_, damping, a_old, b_old, a_new_d, b_new_d,
                            r_d, v_d):
        n = self._n(n)
        ax, bx = self.view(ax_d, 1, "ax")[0], self.view(bx_d, n, "bx").copy()
        assert len(probs) == len(means) == len(vars_) == K, "the mixture tables are HOST arrays of K doubles"
        self._out(a_new_d, b_new_d, n, *orc.gm_prior_message(ax, bx, list(probs), list(means), list(vars_)))

    def _tramp_modulus_lik_bwd(self, ctx, dtype, n, batch, az_d, bz_d, y_d, damping, a_old, b_old, a_new_d, b_new_d, r_d, v_d):
        n = self._n(n)
        az, bz, y = self.view(az_d, 1, "az")[0], self.view(bz_d, 2 * n, "bz").copy().reshape(2, n), self.view(y_d, n, "y").copy()
        a_new, b_new = orc.modulus_lik_message(az, bz, y)
        self._out(a_new_d, b_new_d, 2 * n, a_new, b_new.reshape(-1))

    def _tramp_pwl_channel(self, ctx, dtype, kind, direction, n, batch, az_d, bz_d, ax_d, bx_d, damping, a_old, b_old, a_new_d,
                           b_new_d, r_d, v_d, mask_d):
        n = self._n(n)
        name = {self.L.SGN: "sgn", self.L.ABS: "abs", self.L.RELU: "relu"}[kind]
        az, ax = self.view(az_d, 1, "az")[0], self.view(ax_d, 1, "ax")[0]
        bz, bx = self.view(bz_d, n, "bz"), self.view(bx_d, n, "bx")
        d = "fwd" if direction == self.L.FWD else "bwd"
        self._out(a_new_d, b_new_d, n, *orc.pwl_channel_message(name, d, az, bz.copy(), ax, bx.copy()))

    def _tramp_linear_create(self, ctx, dtype, M, N, rank, Uk_d, Vt_d, sing_d, out):
        M, N, rank = self._n(M), self._n(N), self._n(rank)
        k = min(M, N)
        h = self.next_handle
        self.next_handle += 0x10
        self.handles[h] = dict(M=M, N=N, k=k, rank=rank, Uk=self.view(Uk_d, M * k, "Uk").reshape(M, k),
                               Vt=self.view(Vt_d, k * N, "Vt").reshape(k, N), sing=self.view(sing_d, k, "sing"))
        out._obj.value = h

    def _tramp_linear_destroy(self, ctx, h):
        del self.handles[self._addr(h)]

    def _tramp_linear_message(self, ctx, h, direction, batch, az_d, bz_d, ax_d, bx_d, damping, a_old, b_old, a_new_d, b_new_d, r_d,
                              v_d, reuse):
        lin = self.handles[self._addr(h)]
        M, N, k, rank = lin["M"], lin["N"], lin["k"], lin["rank"]
        assert self._n(batch) == 1 and reuse == 0
        az, ax = self.view(az_d, 1, "az")[0], self.view(ax_d, 1, "ax")[0]
        bz, bx = self.view(bz_d, N, "bz").copy(), self.view(bx_d, M, "bx").copy()
        # the thin form of include/tramp_hip.h (K8) and the spectral scalars of linear_channel.py:58-105
        sg = lin["sing"]
        t, u = lin["Vt"] @ bz, lin["Uk"].T @ bx
        D = az + ax * sg**2
        singular = (sg**2)[:rank]
        n_eff = 0.0 if ax == 0 else (rank / N if az / ax == 0 else np.sum(singular / (az / ax + singular)) / N)
        if direction == self.L.FWD:
            r = lin["Uk"] @ (sg * (t + sg * u) / D)
            v = np.mean(singular) * rank / (M * az) if ax == 0 else n_eff / ((M / N) * ax)
            self._out(a_new_d, b_new_d, M, *orc.ab_new(r, v, ax, bx))
        else:
            r = lin["Vt"].T @ ((t + sg * u) / D) if k == N else bz / az + lin["Vt"].T @ ((1 / D - 1 / az) * t + sg * u / D)
            v = (1 - n_eff) / max(1e-11, az)
            self._out(a_new_d, b_new_d, N, *orc.ab_new(r, v, az, bz))


# ------------------------------------------------------------------------------------------------ models of tools/make_golden.py
def teacher(seed, N, alpha, rho, output, nv):
    rs = np.random.RandomState(seed)
    M = int(alpha * N)
    W = rs.randn(M, N) / np.sqrt(N)
    x = rs.standard_normal(N) * rs.binomial(n=1, size=N, p=rho)
    z = W @ x
    y = z + np.sqrt(nv) * rs.standard_normal(M) if output == "gaussian" else np.sign(z)
    return W, x, y


def build(ref, which):
    from tramp.channels import LinearChannel, ReluChannel
    from tramp.likelihoods import GaussianLikelihood, SgnLikelihood
    from tramp.priors import GaussBernoulliPrior, GaussianPrior
    from tramp.variables import SISOVariable as V
    if which == "sparse_glm":
        W, x, y = teacher(1, 64, 0.5, 0.25, "gaussian", 1e-2)
        return (GaussBernoulliPrior(size=64, rho=0.25) @ V(id="x") @ LinearChannel(W) @ V(id="z")
                @ GaussianLikelihood(y=y, var=1e-2)).to_model(), 4
    if which == "cs1bit":
        W, x, y = teacher(2, 64, 0.7, 0.25, "sgn", None)
        return (GaussBernoulliPrior(size=64, rho=0.25) @ V(id="x") @ LinearChannel(W) @ V(id="z") @ SgnLikelihood(y=y)).to_model(), 4
    if which == "phase_retrieval":
        from tramp.channels import ComplexLinearChannel
        from tramp.likelihoods import ModulusLikelihood
        rs = np.random.RandomState(3)
        N, M = 32, 80
        W = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))) / np.sqrt(2 * N)
        x = rs.standard_normal((2, N)) * rs.binomial(1, 0.6, (2, N))
        y = np.abs(W @ (x[0] + 1j * x[1]))
        return (GaussBernoulliPrior(size=(2, N), rho=0.6) @ V(id="x") @ ComplexLinearChannel(W) @ V(id="z")
                @ ModulusLikelihood(y=y)).to_model(), 4
    if which in ("exp_glm", "positive_glm", "mixture_glm", "binary_glm"):
        from tramp.priors import ExponentialPrior, PositivePrior, GaussianMixturePrior, BinaryPrior
        rs = np.random.RandomState(5)
        N, M = 48, 32
        W = rs.randn(M, N) / np.sqrt(N)
        prior = dict(exp_glm=lambda: ExponentialPrior(size=N, mean=0.7), positive_glm=lambda: PositivePrior(size=N),
                     mixture_glm=lambda: GaussianMixturePrior(size=N, probs=[0.2, 0.5, 0.3], means=[-2.0, 0.0, 1.5],
                                                              vars=[0.3, 0.01, 2.0]),
                     binary_glm=lambda: BinaryPrior(size=N, p_pos=0.3))[which]()
        x = np.abs(rs.standard_normal(N)) if which != "binary_glm" else np.sign(rs.standard_normal(N))
        # (binary: a noisy problem - once tanh saturates, v = 1 - r^2 is round-off and a_new = 1 / v amplifies an ulp of the
        #  incoming message to 1e-5 of the outgoing one, in the reference's own run as in any other)
        nv = 2.0 if which == "binary_glm" else 1e-2
        y = W @ x + np.sqrt(nv) * rs.standard_normal(M)
        return (prior @ V(id="x") @ LinearChannel(W) @ V(id="z") @ GaussianLikelihood(y=y, var=nv)).to_model(), 4
    rs = np.random.RandomState(1)
    N, M1, M2 = 64, 128, 32
    W1, W2 = rs.randn(M1, N) / np.sqrt(N), rs.randn(M2, M1) / np.sqrt(M1)
    x = rs.standard_normal(N)
    y = W2 @ np.maximum(0, W1 @ x) + 0.1 * rs.standard_normal(M2)
    return (GaussianPrior(size=N) @ V(id="x") @ LinearChannel(W1) @ V(id="z1") @ ReluChannel() @ V(id="a1")
            @ LinearChannel(W2) @ V(id="z2") @ GaussianLikelihood(y=y, var=1e-2)).to_model(), 7


def run(ref, model, n_iter, damping, init=None):
    from tramp.algos import ExpectationPropagation
    ep = ExpectationPropagation(model)
    ep.iterate(max_iter=n_iter, damping=damping, initializer=init)
    data = ep.get_variables_data()
    msgs = {(str(s), str(t)): (d["a"], np.array(d["b"], copy=True)) for s, t, d in ep.message_dag.edges(data=True)
            if "a" in d}
    return data, msgs, ep.n_iter


@pytest.mark.parametrize("which", ["sparse_glm", "cs1bit", "two_layer"])
@pytest.mark.parametrize("damping", [None, 0.3])
def test_reference_iterate_runs_through_the_c_abi(ref, which, damping):
    from tramp_amd import compat
    np.random.seed(0)
    model, n_factors = build(ref, which)
    n_iter = 6
    data0, msgs0, it0 = run(ref, model, n_iter, damping)                 # the reference on its own NumPy methods

    lib = FakeLib()
    be = compat.Backend(lib=lib)
    binding = compat.bind_reference(be)
    try:
        model_b, _ = build(ref, which)                                   # LinearChannel.__init__ now uploads its factors
        n_lin = 2 if which == "two_layer" else 1
        assert lib.calls.count("tramp_linear_create") == n_lin
        start = len(lib.calls)
        data1, msgs1, it1 = run(ref, model_b, n_iter, damping)
        compute = [c for c in lib.calls[start:] if c not in ("tramp_buf_alloc", "tramp_buf_free", "tramp_h2d", "tramp_d2h")]
    finally:
        binding.unbind()
    # (a) same run: iteration count, every variable's estimate, every edge's message
    assert it1 == it0
    for vid in data0:
        np.testing.assert_allclose(data1[vid]["r"], data0[vid]["r"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(data1[vid]["v"], data0[vid]["v"], rtol=1e-9)
    assert set(msgs1) == set(msgs0)
    for key in msgs0:
        np.testing.assert_allclose(msgs1[key][0], msgs0[key][0], rtol=1e-9)
        np.testing.assert_allclose(msgs1[key][1], msgs0[key][1], rtol=1e-9, atol=1e-9 * np.abs(msgs0[key][1]).max())
    # (b) exactly the C-ABI calls the header declares for this model, in the reference's schedule: forward sweep
    # (root -> leaf), backward sweep (leaf -> root), tramp/algos/message_passing.py:330-357
    fwd = {"sparse_glm": ["tramp_gb_prior_fwd", "tramp_linear_message"],
           "cs1bit": ["tramp_gb_prior_fwd", "tramp_linear_message"],
           "two_layer": ["tramp_gauss_prior_fwd", "tramp_linear_message", "tramp_pwl_channel", "tramp_linear_message"]}[which]
    bwd = {"sparse_glm": ["tramp_gauss_lik_bwd", "tramp_linear_message"],
           "cs1bit": ["tramp_sgn_lik_bwd", "tramp_linear_message"],
           "two_layer": ["tramp_gauss_lik_bwd", "tramp_linear_message", "tramp_pwl_channel", "tramp_linear_message"]}[which]
    assert compute == (fwd + bwd) * it1, compute[:12]
    # every buffer was released, every call converted to the header's argument types (checked in FakeLib._convert)
    live = sum(nb for _, nb in lib.buffers.values())
    factors = sum(8 * (h["M"] * h["k"] + h["k"] * h["N"] + h["k"]) for h in lib.handles.values())
    assert live == factors, "message buffers leaked"


@pytest.mark.parametrize("which", ["phase_retrieval", "exp_glm", "positive_glm", "mixture_glm", "binary_glm"])
def test_reference_iterate_runs_the_round6_factors_through_the_c_abi(ref, which):
    """the factors added in round 6 under the reference's own scheduler: phase retrieval (GaussBernoulliPrior on a planar
    (2, N) variable -> ComplexLinearChannel -> ModulusLikelihood) and sparse regression behind the Exponential / Positive /
    GaussianMixture / Binary priors"""
    from tramp.algos import ConstantInit, CustomInit
    from tramp_amd import compat
    np.random.seed(0)
    n_iter, damping = 6, 0.3
    if which == "phase_retrieval":
        b0 = np.random.RandomState(9).standard_normal((2, 32))
        init = CustomInit(a_init=[("x", "bwd", 0.1)], b_init=[("x", "bwd", b0)], a=0, b=0)
    else:
        init = ConstantInit(a=1e-3 if which == "exp_glm" else 0, b=0)
    model, _ = build(ref, which)
    with np.errstate(all="ignore"):
        data0, msgs0, it0 = run(ref, model, n_iter, damping, init)
    lib = FakeLib()
    binding = compat.bind_reference(compat.Backend(lib=lib))
    try:
        model_b, _ = build(ref, which)
        assert lib.calls.count("tramp_linear_create") == 1
        start = len(lib.calls)
        with np.errstate(all="ignore"):
            data1, msgs1, it1 = run(ref, model_b, n_iter, damping, init)
        compute = [c for c in lib.calls[start:] if c not in ("tramp_buf_alloc", "tramp_buf_free", "tramp_h2d", "tramp_d2h")]
    finally:
        binding.unbind()
    assert it1 == it0
    for vid in data0:
        np.testing.assert_allclose(data1[vid]["r"], data0[vid]["r"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(data1[vid]["v"], data0[vid]["v"], rtol=1e-8)
    for key in msgs0:
        np.testing.assert_allclose(msgs1[key][0], msgs0[key][0], rtol=1e-8)
        np.testing.assert_allclose(msgs1[key][1], msgs0[key][1], rtol=1e-8, atol=1e-8 * max(np.abs(msgs0[key][1]).max(), 1e-6))
    prior = dict(phase_retrieval="tramp_gb_prior_fwd", exp_glm="tramp_halfline_prior_fwd", positive_glm="tramp_halfline_prior_fwd",
                 mixture_glm="tramp_gm_prior_fwd", binary_glm="tramp_binary_prior_fwd")[which]
    lik = "tramp_modulus_lik_bwd" if which == "phase_retrieval" else "tramp_gauss_lik_bwd"
    assert compute == ([prior, "tramp_linear_message"] + [lik, "tramp_linear_message"]) * it1, compute[:8]
    live = sum(nb for _, nb in lib.buffers.values())
    factors = sum(8 * (h["M"] * h["k"] + h["k"] * h["N"] + h["k"]) for h in lib.handles.values())
    assert live == factors, "message buffers leaked"


def test_unbind_restores_the_reference(ref):
    from tramp.priors import GaussBernoulliPrior
    from tramp_amd import compat
    before = GaussBernoulliPrior.__dict__.get("compute_forward_message")
    lib = FakeLib()
    binding = compat.bind_reference(compat.Backend(lib=lib))
    assert GaussBernoulliPrior.__dict__.get("compute_forward_message") is not before
    p = GaussBernoulliPrior(size=6, rho=0.1)
    b = np.array([-3, -.5, 0, .25, 2, 6.])
    a_new, b_new = p.compute_forward_message(1.3, b)                     # SURVEY 8c known answer, through the binding
    binding.unbind()
    assert GaussBernoulliPrior.__dict__.get("compute_forward_message") is before
    assert a_new == pytest.approx(3.4794334296687266, rel=1e-12)
    np.testing.assert_allclose(b_new, [0.8718318506389249, 0.4253971012695395, 0, -0.21408542134028413, -1.3816196819918467,
                                       6.400522480051279], rtol=1e-12)
    a_ref, b_ref = p.compute_forward_message(1.3, b)                     # the reference's own method again
    assert a_ref == pytest.approx(a_new, rel=1e-12)
    assert "tramp_gb_prior_fwd" in lib.calls


def test_stand_in_rejects_a_wrong_call(ref):
    "the recording library is strict: a call with a missing argument or a Python float where the header has a pointer fails"
    lib = FakeLib()
    with pytest.raises(AssertionError):
        lib.tramp_gb_prior_fwd(C.c_void_p(1), 1, C.c_size_t(4), C.c_size_t(1))
    with pytest.raises(AssertionError):
        lib.tramp_gauss_prior_fwd(C.c_void_p(1), 1, C.c_size_t(4), C.c_size_t(1), 0.0, 1.0, 3.5, None)
