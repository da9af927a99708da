"""GPU parity of the device setup path (tramp_linear_factorize / tramp_eigh) against the goldens
captured from the reference's LinearChannel.__init__ (tramp/channels/linear/linear_channel.py:8-15,
30-46: np.linalg.matrix_rank + full SVD) and against NumPy.

Three configurations of the rounds 2-5 eigensolver (TRAMP_EIG_METHOD=split) are exercised on the same cases: rocSOLVER
alone, one spectral divide-and-conquer split (TRAMP_EIG_CAP below the matrix size), and a cap small enough to force
recursion; then the round-6 default, the tridiagonal path (one stedc call, and the Cuppen merge BASELINE configs[3]
takes at k = 50000, forced at small orders) - see the end of the file.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tramp_amd
    return tramp_amd


class eig_cap:
    "TRAMP_EIG_CAP / TRAMP_EIG_METHOD are read (getenv) by every tramp_linear_factorize / tramp_eigh call."

    def __init__(self, cap):
        self.cap = cap

    def __enter__(self):
        self.old_method = os.environ.get("TRAMP_EIG_METHOD")
        os.environ["TRAMP_EIG_METHOD"] = "split"
        self.old = os.environ.get("TRAMP_EIG_CAP")
        if self.cap is None:
            os.environ.pop("TRAMP_EIG_CAP", None)
        else:
            os.environ["TRAMP_EIG_CAP"] = str(self.cap)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("TRAMP_EIG_CAP", None)
        else:
            os.environ["TRAMP_EIG_CAP"] = self.old
        if self.old_method is None:
            os.environ.pop("TRAMP_EIG_METHOD", None)
        else:
            os.environ["TRAMP_EIG_METHOD"] = self.old_method


def _W(case):
    if "W" in case:
        return case["W"]
    rs = np.random.RandomState(case["seed"])
    return rs.randn(case["M"], case["N"]) / np.sqrt(case["N"])


def _eigh(T, ctx, G):
    n = G.shape[0]
    A = ctx.array(np.ascontiguousarray(G), np.float64)
    w = ctx.empty((n,), np.float64)
    ctx._call("tramp_eigh", n, A.ptr, w.ptr)
    V = A.get().T                      # device result is column-major: row-major view is V^T
    return w.get(), np.ascontiguousarray(V)


@pytest.mark.parametrize("cap", [None, 24, 6])
def test_device_factorization_golden(T, ctx, golden_linear, cap):
    "spectrum, rank, vz(az=0) and the posteriors of the reference, with the factors computed on the device"
    for case in golden_linear:
        W = _W(case)
        M, N = W.shape
        k = min(M, N)
        with eig_cap(cap):
            ch = T.LinearChannel(W, ctx=ctx, svd="device")
        assert ch.rank == int(case["rank"]), (M, N, cap)
        ref = np.asarray(case["spectrum"])
        np.testing.assert_allclose(ch.spectrum, ref, rtol=0, atol=1e-12 * ref.max())
        # thin factors reproduce W and are orthonormal on the rank
        Uk, s, Vt = ch._Uk.get(), ch._sing.get(), ch._Vt.get()
        r = ch.rank
        np.testing.assert_allclose((Uk[:, :r] * s[:r]) @ Vt[:r], W, rtol=0, atol=1e-12 * s[0])
        np.testing.assert_allclose(Vt[:r] @ Vt[:r].T, np.eye(r), rtol=0, atol=1e-11)
        np.testing.assert_allclose(Uk[:, :r].T @ Uk[:, :r], np.eye(r), rtol=0, atol=1e-11)
        bz, bx = case["bz"], case["bx"]
        for c in case["cases"]:
            az, ax = c["az"], c["ax"]
            rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
            rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
            cancel = 4e-16 * np.abs(bz).max() / max(az, 1e-11) * N
            np.testing.assert_allclose(rz, c["rz"], rtol=1e-9, atol=1e-9 * np.abs(c["rz"]).max())
            np.testing.assert_allclose(rx, c["rx"], rtol=1e-9, atol=1e-9 * np.abs(c["rx"]).max() + 10 * cancel)
            np.testing.assert_allclose(vx, c["vx"], rtol=1e-9)
            np.testing.assert_allclose(vz, c["vz"], rtol=1e-9, atol=64 * 2.3e-16 / max(az, 1e-11))
        if "vz_az0" in case:
            v, _, _ = ch.device_variance(T._lib.BWD, 0.0, 0.7)
            np.testing.assert_allclose(v, case["vz_az0"], rtol=1e-9)


@pytest.mark.parametrize("n,cap", [(257, 100), (700, 256), (1500, None)])
def test_eigh_against_numpy(T, ctx, n, cap):
    "tramp_eigh on a Marchenko-Pastur Gram matrix: eigenvalues, residual and orthogonality"
    rs = np.random.RandomState(n)
    Wm = rs.randn(n, 2 * n) / np.sqrt(2 * n)
    G = Wm @ Wm.T
    with eig_cap(cap):
        w, V = _eigh(T, ctx, G)
    wr = np.linalg.eigvalsh(G)
    nG = wr.max()
    assert np.all(np.diff(w) >= -1e-13 * nG)
    np.testing.assert_allclose(w, wr, rtol=0, atol=1e-13 * nG)
    assert np.abs(G @ V - V * w).max() < 1e-13 * nG
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12


def test_eigh_split_point_on_an_eigenvalue(T, ctx):
    """an eigenvalue 1e-12 away from the split point sigma = trace/n: the residual-certified bound has
    to take over from the predicted one (factorize.hip, eigh_split)"""
    n = 300
    rs = np.random.RandomState(3)
    Q, _ = np.linalg.qr(rs.randn(n, n))
    lam = np.linspace(-1.0, 1.0, n)
    lam[n // 2] = 1e-12
    lam -= lam.mean() - 0.0                 # trace/n stays ~0: sigma sits (almost) on lam[n/2]
    G = (Q * lam) @ Q.T
    G = 0.5 * (G + G.T) + 2.0 * np.eye(n)
    with eig_cap(200):
        w, V = _eigh(T, ctx, G)
    wr = np.linalg.eigvalsh(G)
    np.testing.assert_allclose(w, wr, rtol=0, atol=1e-13 * wr.max())
    assert np.abs(G @ V - V * w).max() < 1e-13 * wr.max()
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12


@pytest.mark.parametrize("shape", [(60, 90), (90, 60)])
@pytest.mark.parametrize("cap", [None, 32])
def test_factorization_ill_conditioned(T, ctx, shape, cap):
    """singular values over 12 decades: the Gram route alone stops at 1e-8 sigma_max, the second
    level (Gram matrix of the trailing block's own rows) recovers them to LAPACK's absolute
    accuracy, and the rank agrees with np.linalg.matrix_rank"""
    M, N = shape
    k = min(M, N)
    rs = np.random.RandomState(11)
    U, _ = np.linalg.qr(rs.randn(M, k))
    V, _ = np.linalg.qr(rs.randn(N, k))
    s = np.logspace(0, -12, k)
    s[-3:] = 0.0                                            # and an exact null space
    W = (U * s) @ V.T
    with eig_cap(cap):
        ch = T.LinearChannel(W, ctx=ctx, svd="device")
    s_ref = np.linalg.svd(W, compute_uv=False)
    assert ch.rank == np.linalg.matrix_rank(W), (ch._sing.get()[-6:], s_ref[-6:])
    # what the message kernels use is the spectrum sigma^2: absolute accuracy eps * sigma_max^2 (a Gram
    # matrix gives d(sigma) = d(lambda) / (2 sigma), so sigma itself is compared on that scale)
    sd = ch._sing.get()
    np.testing.assert_allclose(sd**2, s_ref**2, rtol=0, atol=1e-13)
    assert np.all(np.abs(sd - s_ref) <= 1e-13 / np.maximum(2 * s_ref, 1e-8) + 1e-13)
    # the resolvent the messages use, against the oracle on LAPACK's factors
    lin = orc.LinearOracle(W)
    bz, bx = rs.standard_normal(N), rs.standard_normal(M)
    for az, ax in ((1.3, 0.7), (1e-3, 10.0), (5.0, 1e4)):
        rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
        rz_ref, vz_ref = lin.backward_mean(az, bz, ax, bx), lin.backward_variance(az, ax)
        np.testing.assert_allclose(rz, rz_ref, rtol=1e-7, atol=1e-8 * np.abs(rz_ref).max())
        np.testing.assert_allclose(vz, vz_ref, rtol=1e-8)


def test_factorization_f32_inputs(T, ctx):
    rs = np.random.RandomState(5)
    W = (rs.randn(64, 160) / np.sqrt(160)).astype(np.float32)
    with eig_cap(40):
        ch = T.LinearChannel(W, dtype=np.float32, ctx=ctx, svd="device")
    s_ref = np.linalg.svd(W.astype(np.float64), compute_uv=False)
    np.testing.assert_allclose(ch._sing.get(), s_ref, rtol=1e-6)
    assert ch.rank == 64
    Vt = ch._Vt.get().astype(np.float64)
    np.testing.assert_allclose(Vt @ Vt.T, np.eye(64), atol=5e-6)


# ------------------------------------------------------------------------------------------------------------------
# round 6: the tridiagonal path (csrc/tridiag_eig.h) - own blocked Householder tridiagonalisation, rocSOLVER stedc on the
# tridiagonal matrix, one Cuppen merge of two halves where stedc's 32-bit indexing ends (forced at small orders with
# TRAMP_TD_MERGE_ABOVE), blocked back-transformation.  Same bar as the other two eigensolver configurations.
class env_set:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _tridiag_matrix(kind, n):
    rs = np.random.RandomState(n + len(kind))
    if kind == "gram":
        Wm = rs.randn(n, 2 * n) / np.sqrt(2 * n)
        return Wm @ Wm.T
    if kind == "deficient":                      # half the spectrum is an n/2-fold zero eigenvalue: deflation
        Wm = rs.randn(n, n // 2) / np.sqrt(n)
        return Wm @ Wm.T
    if kind == "indefinite":
        B = rs.randn(n, n)
        return 0.5 * (B + B.T)
    if kind == "identity":
        return np.eye(n)
    if kind == "tridiagonal":                    # already tridiagonal: every reflector is the identity (tau = 0)
        e = rs.randn(n - 1)
        return np.diag(rs.rand(n) + 2.0) + np.diag(e, 1) + np.diag(e, -1)
    Q, _ = np.linalg.qr(rs.randn(n, n))
    if kind == "clustered":                      # pairs of eigenvalues 1e-13 apart: Givens deflation in the merge
        lam = np.repeat([1.0, 2.0, 2.0 + 1e-13, 5.0], n // 4 + 1)[:n]
    else:                                        # "graded": twelve decades
        lam = np.logspace(-12, 0, n)
    G = (Q * lam) @ Q.T
    return 0.5 * (G + G.T)


MERGES = {"stedc": dict(TRAMP_TD_MERGE_ABOVE=None, TRAMP_TD_MERGE_DENSE=None),            # one stedc call
          "merge": dict(TRAMP_TD_MERGE_ABOVE=64, TRAMP_TD_MERGE_DENSE=None),              # two halves + the merge (per-half basis
          "merge-dense": dict(TRAMP_TD_MERGE_ABOVE=64, TRAMP_TD_MERGE_DENSE=1)}           # unless a rotation crosses them) / dense


@pytest.mark.parametrize("merge", list(MERGES))
@pytest.mark.parametrize("kind,n", [("gram", 64), ("gram", 257), ("gram", 1031), ("deficient", 700), ("indefinite", 513),
                                    ("identity", 300), ("tridiagonal", 400), ("clustered", 512), ("graded", 600)])
def test_eigh_tridiagonal_path(T, ctx, kind, n, merge):
    G = _tridiag_matrix(kind, n)
    with env_set(TRAMP_EIG_METHOD="tridiag", **MERGES[merge]):
        w, V = _eigh(T, ctx, G)
    wr = np.linalg.eigvalsh(G)
    nG = np.abs(wr).max()
    assert np.all(np.diff(w) >= 0.0)
    np.testing.assert_allclose(w, wr, rtol=0, atol=1e-13 * nG)
    assert np.abs(G @ V - V * w).max() < 2e-13 * nG
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12


@pytest.mark.parametrize("merge", list(MERGES))
def test_device_factorization_golden_tridiagonal_path(T, ctx, golden_linear, merge):
    "the golden factorisations of the reference's LinearChannel.__init__ through the tridiagonal path"
    with env_set(TRAMP_EIG_METHOD="tridiag", **MERGES[merge]):
        test_device_factorization_golden(T, ctx, golden_linear, None)


def test_eigh_tridiagonal_path_large(T, ctx):
    "n = 3000 (a dozen panels, ragged last tile) with and without the merge: residual and orthogonality on the device result"
    n = 3000
    G = _tridiag_matrix("gram", n)
    wr = np.linalg.eigvalsh(G)
    for merge_above, dense, panel in ((None, None, None), (1024, None, 256), (1024, 1, 1000)):
        with env_set(TRAMP_EIG_METHOD="tridiag", TRAMP_TD_MERGE_ABOVE=merge_above, TRAMP_TD_MERGE_DENSE=dense, TRAMP_TD_BT_PANEL=panel):
            w, V = _eigh(T, ctx, G)
        np.testing.assert_allclose(w, wr, rtol=0, atol=1e-13 * wr.max())
        assert np.abs(G @ V - V * w).max() < 2e-13 * wr.max()
        assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12
