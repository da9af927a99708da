#!/usr/bin/env python3
"""NumPy prototype of the round-6 eigensolver of the set-up path (csrc/tridiag.hip): blocked Householder tridiagonalisation
(LAPACK dsytrd / dlatrd, lower triangle), one Cuppen merge of two half problems (dlaed1-4: deflation, secular equation
by bisection about the nearer pole, Loewner-recomputed z, eigenvectors, one gemm per half) and the blocked
back-transformation (compact WY).  Every array operation here is one kernel / library call there, in the same order.

    python tools/proto_tridiag_eig.py [n]
"""
import sys

import numpy as np
from scipy.linalg import eigh_tridiagonal

EPS = np.finfo(float).eps


# ------------------------------------------------------------------ dsytrd (lower), blocked
def sytrd_lower(A, nb=32):
    """A: symmetric n x n (only the lower triangle is read / updated).  Returns d, e, tau; the reflector vectors are left
    below the sub-diagonal of A (v[0] = 1 implicit), as LAPACK."""
    n = A.shape[0]
    d, e, tau = np.zeros(n), np.zeros(max(n - 1, 0)), np.zeros(max(n - 1, 0))
    i = 0
    while i < n - 1:
        pb = min(nb, n - 1 - i)                    # reflectors in this panel
        m = n - i
        At = A[i:, i:]                             # trailing matrix (view)
        W = np.zeros((m, pb))
        for j in range(pb):
            # column j of the trailing matrix, updated by the panel's earlier reflectors
            if j:
                At[j:, j] -= At[j:, :j] @ W[j, :j] + W[j:, :j] @ At[j, :j]
            d[i + j] = At[j, j]
            # reflector annihilating At[j+2:, j]
            alpha = At[j + 1, j]
            x = At[j + 2:, j]
            xn = np.linalg.norm(x)
            if xn == 0.0:
                t, beta = 0.0, alpha
            else:
                beta = -np.copysign(np.hypot(alpha, xn), alpha)
                t = (beta - alpha) / beta
                At[j + 2:, j] = x / (alpha - beta)
            e[i + j], tau[i + j] = beta, t
            At[j + 1, j] = 1.0
            v = At[j + 1:, j]
            # w = tau (A22 v - V (W^T v) - W (V^T v)), then w += -(tau/2)(w.v) v        (A22: the ORIGINAL trailing block)
            L = At[j + 1:, j + 1:]
            w = np.tril(L) @ v + np.tril(L, -1).T @ v                        # symv on the lower triangle
            if j:
                w -= At[j + 1:, :j] @ (W[j + 1:, :j].T @ v) + W[j + 1:, :j] @ (At[j + 1:, :j].T @ v)
            w *= t
            w += (-0.5 * t * (w @ v)) * v
            W[j + 1:, j] = w
        # trailing update A22 -= V W^T + W V^T  (lower triangle), V = At[pb:, :pb] with its unit entries in place
        V = At[pb:, :pb]
        At[pb:, pb:] -= np.tril(V @ W[pb:, :].T + W[pb:, :] @ V.T)
        for j in range(pb):
            At[j + 1, j] = e[i + j]                # restore the sub-diagonal
        i += pb
    d[n - 1] = A[n - 1, n - 1]
    return d, e, tau


def backtransform(A, tau, C, nb=32):
    "C <- Q C with Q = H(0) H(1) ... H(n-2) (reflectors below the sub-diagonal of A): blocked, last panel first"
    n = A.shape[0]
    starts = list(range(0, n - 1, nb))
    for i in reversed(starts):
        pb = min(nb, n - 1 - i)
        V = np.tril(A[i + 1:, i:i + pb], -1) + np.eye(n - i - 1, pb)     # unit lower trapezoidal
        # T of the compact WY form (dlarft, forward columnwise)
        S = V.T @ V
        T = np.zeros((pb, pb))
        for j in range(pb):
            T[j, j] = tau[i + j]
            if j:
                T[:j, j] = -tau[i + j] * (T[:j, :j] @ S[:j, j])
        C[i + 1:, :] -= V @ (T @ (V.T @ C[i + 1:, :]))
    return C


# ------------------------------------------------------------------ one Cuppen merge
def secular_roots(dk, zk, rho, iters=200):
    """roots of 1 + rho sum z_j^2 / (d_j - lam) (rho > 0, d ascending, no z_j = 0), each as (origin index, mu) with
    lam = d[origin] + mu: bisection in mu about the nearer pole (geometric while the bracket spans decades)."""
    k = dk.size
    z2 = zk * zk
    org = np.zeros(k, dtype=int)
    mu = np.zeros(k)
    for i in range(k):
        hi_pole = dk[i + 1] if i + 1 < k else dk[k - 1] + rho * z2.sum()
        gap = hi_pole - dk[i]
        # sign of f at the midpoint decides the origin (dlaed4)
        mid = 0.5 * gap
        fm = 1.0 + rho * np.sum(z2 / ((dk - dk[i]) - mid))
        if fm > 0 or i + 1 == k:
            o, lo, hi = i, 0.0, (mid if i + 1 < k else gap)          # root in (d_i, d_i + mid]
            if i + 1 == k and fm <= 0:
                lo, hi = mid, gap
        else:
            o, lo, hi = i + 1, -mid, 0.0                              # root in [d_{i+1} - mid, d_{i+1})
        delta = dk - dk[o]
        for _ in range(iters):
            if lo != 0.0 and hi != 0.0 and (hi / lo > 4.0 or lo / hi > 4.0):
                m_ = np.sign(lo) * np.sqrt(lo * hi)
            elif lo == 0.0 and hi != 0.0 and _ < 60:
                m_ = hi * 2.0 ** -8 if _ else hi * 0.5
            elif hi == 0.0 and lo != 0.0 and _ < 60:
                m_ = lo * 2.0 ** -8 if _ else lo * 0.5
            else:
                m_ = 0.5 * (lo + hi)
            if m_ == lo or m_ == hi:
                break
            f = 1.0 + rho * np.sum(z2 / (delta - m_))
            if f > 0:
                hi = m_
            else:
                lo = m_
        org[i], mu[i] = o, 0.5 * (lo + hi) if (lo != 0.0 and hi != 0.0) else (hi if lo == 0.0 else lo)
    return org, mu


def merge(D1, Q1, D2, Q2, beta):
    """eigen-decomposition of [[T1, beta e e^T], [., T2]] from those of T1' = T1 - |beta| e_last e_last^T and
    T2' = T2 - |beta| e_first e_first^T (dlaed1): returns ascending eigenvalues and the eigenvector matrix."""
    n1, n2 = D1.size, D2.size
    n = n1 + n2
    s = 1.0 if beta >= 0 else -1.0
    z = np.concatenate([Q1[-1, :], s * Q2[0, :]]) / np.sqrt(2.0)
    rho = 2.0 * abs(beta)
    d = np.concatenate([D1, D2])
    Q = np.zeros((n, n))
    Q[:n1, :n1], Q[n1:, n1:] = Q1, Q2
    order = np.argsort(d, kind="stable")
    d, z, Q = d[order], z[order], Q[:, order]
    # ---- deflation (dlaed2): negligible z_j, then pairs of close d's by a Givens rotation
    tol = 8.0 * EPS * max(np.abs(d).max(), np.abs(z).max())
    keep = rho * np.abs(z) > tol
    idx = np.flatnonzero(keep)
    live = []
    for j in idx:
        if live:
            p = live[-1]
            c_, s_ = z[j], z[p]
            t_ = np.hypot(c_, s_)
            if abs((d[j] - d[p]) * (c_ / t_) * (s_ / t_)) <= tol:
                # rotate (p, j) so that z[p] = 0: p deflates with its rotated eigenvalue
                c_, s_ = c_ / t_, -s_ / t_
                qp, qj = Q[:, p].copy(), Q[:, j].copy()
                Q[:, p], Q[:, j] = c_ * qp + s_ * qj, c_ * qj - s_ * qp
                dp, dj = d[p], d[j]
                d[p], d[j] = dp * c_ * c_ + dj * s_ * s_, dp * s_ * s_ + dj * c_ * c_
                z[j], z[p] = t_, 0.0
                keep[p] = False
                live[-1] = j
                continue
        live.append(j)
    live = np.array(live, dtype=int)
    lam = d.copy()
    out = Q.copy()
    k = live.size
    if k:
        dk, zk = d[live], z[live]
        o = np.argsort(dk, kind="stable")
        dk, zk, live = dk[o], zk[o], live[o]
        org, mu = secular_roots(dk, zk, rho)
        # differences d_j - lam_i without cancellation: (d_j - d_org(i)) - mu_i
        diff = (dk[:, None] - dk[org][None, :]) - mu[None, :]              # [j, i]
        # Loewner: zhat_j^2 = prod_i (lam_i - d_j) / (rho prod_{i != j} (d_i - d_j))   (Gu & Eisenstat), signs from z
        num = -diff                                                          # lam_i - d_j
        den = dk[None, :] - dk[:, None]                                      # d_i - d_j
        np.fill_diagonal(den, 1.0)
        zhat = np.sqrt(np.abs(np.prod(num / den, axis=1) / rho)) * np.sign(zk)
        V = zhat[:, None] / diff
        V /= np.linalg.norm(V, axis=0)[None, :]
        out[:, live] = Q[:, live] @ V
        lam[live] = dk[org] + mu
    order = np.argsort(lam, kind="stable")
    return lam[order], out[:, order]


def eigh_tridiag_merge(d, e):
    n = d.size
    m = n // 2
    beta = e[m - 1]
    d1, d2 = d[:m].copy(), d[m:].copy()
    d1[-1] -= abs(beta)
    d2[0] -= abs(beta)
    D1, Q1 = eigh_tridiagonal(d1, e[:m - 1], lapack_driver='stev')
    D2, Q2 = eigh_tridiagonal(d2, e[m:], lapack_driver='stev')
    return merge(D1, Q1, D2, Q2, beta)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    rs = np.random.RandomState(1)
    for kind in ("gram", "gram-rank-deficient", "clustered"):
        if kind == "gram":
            W = rs.randn(n, 2 * n) / np.sqrt(2 * n)
            G = W @ W.T
        elif kind == "gram-rank-deficient":
            W = rs.randn(n, n // 2) / np.sqrt(n)
            G = W @ W.T
        else:
            Qr, _ = np.linalg.qr(rs.randn(n, n))
            G = (Qr * np.repeat([1.0, 2.0, 2.0 + 1e-13, 5.0], n // 4 + 1)[:n]) @ Qr.T
            G = 0.5 * (G + G.T)
        A = G.copy()
        d, e, tau = sytrd_lower(A, nb=24)
        lam, QT = eigh_tridiag_merge(d, e)
        ref = np.linalg.eigvalsh(G)
        Q = backtransform(A, tau, QT.copy(), nb=24)
        print(f"{kind:22s} n={n}: |lam - ref| / |ref|max = {np.abs(lam - ref).max() / np.abs(ref).max():.2e}   "
              f"|G Q - Q lam| / |G| = {np.abs(G @ Q - Q * lam).max() / np.abs(G).max():.2e}   "
              f"|Q^T Q - I| = {np.abs(Q.T @ Q - np.eye(n)).max():.2e}")


if __name__ == "__main__":
    main()
