"""NumPy prototype of the spectral divide-and-conquer eigensolver that factorize.hip runs on the
device for k = min(M, N) above rocSOLVER's 32-bit cap (k > 46340).

    eigh(G):  sigma = trace(G)/k;  S = sign(G - sigma I) by the dynamically weighted Halley
              iteration in its Cholesky form (only syrk / potrf / trsm, capped weight);
              P = (I + S)/2 projects on the eigenspace above sigma, r = trace(P);
              V1 = CholQR2(P[:, :r]), V2 = CholQR2((I - P)[:, r:]);
              solve V1^T G V1 and V2^T G V2 with the library eigensolver (each <= cap), rotate back.

The prototype is how the iteration parameters (cap on the Halley weight) were chosen: it reports the
backward error |G V - V diag(w)| / |G| and the orthogonality defect for a Marchenko-Pastur Gram matrix.
Run:  python tools/proto_eig_dc.py [k] [cap]
"""
import sys
import numpy as np
import scipy.linalg as sl


def halley_params(l):
    l2 = l * l
    dd = np.cbrt(4.0 * (1.0 / l2 - 1.0) / l2)
    sqd = np.sqrt(1.0 + dd)
    a = sqd + np.sqrt(2.0 - dd + 2.0 * (2.0 - l2) / (l2 * sqd))
    b = (a - 1.0) ** 2 / 4.0
    c = a + b - 1.0
    return a, b, c


def sign_cholesky(A, l0=1e-8, l_floor=1e-3, max_iter=60, log=None):
    """sign(A) of a symmetric matrix: X <- (b/c) X + (a - b/c) X (I + c X^2)^-1, exactly the control
    flow of eigh_split in tramp_amd/csrc/factorize.hip: the weights follow the predicted lower bound
    l_pred (floored at l_floor so that c <= 1.6e4); once the prediction says "converged" but the
    residual e = |X^2 - I|_F does not (an eigenvalue closer to the split point than l0 assumed), the
    bound the residual certifies, sqrt(1 - e), takes over."""
    k = A.shape[0]
    X = A / np.linalg.norm(A, "fro")            # >= spectral norm
    lp = l0
    for it in range(max_iter):
        Z0 = X @ X.T
        e = np.linalg.norm(Z0 - np.eye(k), "fro")
        if log is not None:
            log.append((it, e))
        if e < 1e-13 * k + 1e-12:
            return X
        if lp < 1 - 1e-8:
            l = max(l_floor, lp)
        else:
            l = max(l_floor, np.sqrt(1 - e) if e < 1 else 0.0)
        l = min(l, 1 - 1e-12)
        a, b, c = halley_params(l)
        lp = min(1.0, lp * (a + b * lp * lp) / (1 + c * lp * lp))
        L = np.linalg.cholesky(c * Z0 + np.eye(k))
        Y = sl.cho_solve((L, True), X)
        X = (b / c) * X + (a - b / c) * Y
        X = 0.5 * (X + X.T)
    raise RuntimeError("matrix sign iteration did not converge")


def cholqr2(Y, G0=None):
    G = Y.T @ Y if G0 is None else G0
    R = np.linalg.cholesky(G).T
    Q = sl.solve_triangular(R.T, Y.T, lower=True).T
    G = Q.T @ Q
    R = np.linalg.cholesky(G).T
    return sl.solve_triangular(R.T, Q.T, lower=True).T


def eigh_dc(G, cap, depth=0, stats=None):
    k = G.shape[0]
    if k <= cap:
        return np.linalg.eigh(G)
    sigma = np.trace(G) / k
    log = []
    S = sign_cholesky(G - sigma * np.eye(k), log=log)
    P = 0.5 * (np.eye(k) + S)
    r = int(round(np.trace(P)))
    if stats is not None:
        stats.append((depth, k, r, len(log)))
    V1 = cholqr2(P[:, :r], P[:r, :r])
    V2 = cholqr2((np.eye(k) - P)[:, r:], (np.eye(k) - P)[r:, r:])
    w1, Q1 = eigh_dc(V1.T @ G @ V1, cap, depth + 1, stats)
    w2, Q2 = eigh_dc(V2.T @ G @ V2, cap, depth + 1, stats)
    w = np.concatenate([w2, w1])              # ascending: below sigma first
    V = np.concatenate([V2 @ Q2, V1 @ Q1], axis=1)
    return w, V


if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else (k * 2) // 3
    rs = np.random.RandomState(0)
    W = rs.randn(k, 2 * k) / np.sqrt(2 * k)
    G = W @ W.T
    stats = []
    w, V = eigh_dc(G, cap, stats=stats)
    wr = np.linalg.eigvalsh(G)
    nG = np.linalg.norm(G, 2)
    print("splits (depth, k, r, halley iterations):", stats)
    print("eigenvalue error      %.2e" % (np.abs(w - wr).max() / nG))
    print("backward error        %.2e" % (np.abs(G @ V - V * w).max() / nG))
    print("orthogonality defect  %.2e" % np.abs(V.T @ V - np.eye(k)).max())
