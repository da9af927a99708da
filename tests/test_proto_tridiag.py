"""The NumPy prototype of the set-up path's eigensolver (tools/proto_tridiag_eig.py: blocked dsytrd on the lower triangle, one
Cuppen merge with deflation / bisection about the nearer pole / Loewner's vector, compact-WY back-transformation) - the same
operations in the same order as csrc/tridiag_eig.h, runnable without a GPU.  Host logic only; the device code is checked in
tests/test_gpu_factorize.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import proto_tridiag_eig as proto  # noqa: E402


def _matrix(kind, n, rs):
    if kind == "gram":
        W = rs.randn(n, 2 * n) / np.sqrt(2 * n)
        return W @ W.T
    if kind == "deficient":
        W = rs.randn(n, n // 2) / np.sqrt(n)
        return W @ W.T
    Q, _ = np.linalg.qr(rs.randn(n, n))
    G = (Q * np.repeat([1.0, 2.0, 2.0 + 1e-13, 5.0], n // 4 + 1)[:n]) @ Q.T
    return 0.5 * (G + G.T)


@pytest.mark.parametrize("kind", ["gram", "deficient", "clustered"])
def test_prototype_eigensolver(kind):
    n = 150
    G = _matrix(kind, n, np.random.RandomState(2))
    A = G.copy()
    d, e, tau = proto.sytrd_lower(A, nb=16)
    # the tridiagonal matrix has G's spectrum
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    ref = np.linalg.eigvalsh(G)
    np.testing.assert_allclose(np.linalg.eigvalsh(T), ref, atol=1e-13 * np.abs(ref).max())
    lam, QT = proto.eigh_tridiag_merge(d, e)
    np.testing.assert_allclose(lam, ref, atol=1e-13 * np.abs(ref).max())
    Q = proto.backtransform(A, tau, QT.copy(), nb=16)
    assert np.abs(G @ Q - Q * lam).max() < 1e-13 * np.abs(G).max()
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-12


def test_prototype_secular_roots_interlace():
    rs = np.random.RandomState(4)
    k = 40
    d = np.sort(rs.rand(k))
    z = rs.randn(k) / np.sqrt(k)
    rho = 0.7
    org, mu = proto.secular_roots(d, z, rho)
    lam = d[org] + mu
    assert np.all(lam > d) and np.all(lam[:-1] < d[1:]) and lam[-1] < d[-1] + rho * (z @ z) * (1 + 1e-12)
    np.testing.assert_allclose(lam, np.linalg.eigvalsh(np.diag(d) + rho * np.outer(z, z)), atol=1e-14)
