"""tramp_eigh on a test matrix of order n: time, residual, orthogonality.
    TRAMP_EIG_METHOD=tridiag|split TRAMP_EIG_VERBOSE=1 python tools/probes/probe_eigh.py [--kind gram|deficient|clustered|identity] n [n ...]
gram: Marchenko-Pastur Gram matrix W W^T, W n x 2n (drawn on the device); deficient: W n x n/2 (half the spectrum zero);
clustered: eigenvalues {1, 2, 2 + 1e-13, 5} in equal shares; identity: the unit matrix."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
args = sys.argv[1:]
kind = "gram"
if args and args[0] == "--kind":
    kind = args[1]
    args = args[2:]
for n in [int(a) for a in args]:
    G = ctx.empty((n, n), np.float64)
    Gh = None
    if kind in ("gram", "deficient"):
        k = (2 * n if n <= 32768 else n) if kind == "gram" else n // 2
        W = ctx.randn((n, k), np.float64, seed=5, scale=1.0 / np.sqrt(k))
        if n <= 8192:
            Wh = W.get()
            Gh = Wh @ Wh.T
            G.set(Gh)
        else:
            for r0 in range(0, n, 4096):                         # rows r0 .. of G = W[r0:] W^T, a dense pass per chunk of instances
                nb = min(4096, n - r0)
                ctx._call("tramp_matvec_nt", T._lib.F64, 2, n, k, nb, W.ptr, W.ptr + 8 * k * r0, G.ptr + 8 * n * r0)
        del W
    elif kind == "clustered":
        rs = np.random.RandomState(1)
        Qr, _ = np.linalg.qr(rs.randn(n, n))
        Gh = (Qr * np.repeat([1.0, 2.0, 2.0 + 1e-13, 5.0], n // 4 + 1)[:n]) @ Qr.T
        Gh = 0.5 * (Gh + Gh.T)
        G.set(Gh)
    else:
        Gh = np.eye(n)
        G.set(Gh)
    A = ctx.empty((n, n), np.float64)
    A.copy_from(G)
    w = ctx.empty((n,), np.float64)
    ctx.sync()
    t0 = time.perf_counter()
    ctx._call("tramp_eigh", n, A.ptr, w.ptr)
    ctx.sync()
    dt = time.perf_counter() - t0
    line = f"{kind} n = {n}: {dt:8.3f} s"
    if Gh is not None:
        Q = A.get().T                     # column-major eigenvectors
        lam = w.get()
        res = np.abs(Gh @ Q - Q * lam).max() / np.abs(Gh).max()
        orth = np.abs(Q.T @ Q - np.eye(n)).max()
        ref = np.linalg.eigvalsh(Gh)
        line += f"   |GQ - Q lam| / |G| = {res:.2e}   |Q^T Q - I| = {orth:.2e}   |lam - ref| / |ref| = {np.abs(lam - ref).max() / ref.max():.2e}"
    else:
        # device-side check on a sample of 64 eigenpairs (spread over the spectrum): residual through the library's dense pass
        # on the kept copy of G, orthogonality of the sample against itself
        idx = np.unique(np.linspace(0, n - 1, 64).astype(int))
        from tramp_amd.device import DeviceArray
        Qs = np.stack([DeviceArray(ctx, (n,), np.float64, ptr=A.ptr + 8 * n * int(i), owner=False).get() for i in idx])  # row i of the row-major view = eigenvector i
        lam = w.get()
        X = ctx.array(Qs, np.float64)
        Y = ctx.empty(Qs.shape, np.float64)
        ctx._call("tramp_matvec_nt", T._lib.F64, 2, n, n, len(idx), G.ptr, X.ptr, Y.ptr)
        R = Y.get() - Qs * lam[idx][:, None]
        line += (f"   sample of {len(idx)} eigenpairs: |G q - lam q| / |G| = {np.abs(R).max() / lam.max():.2e}   "
                 f"|Qs Qs^T - I| = {np.abs(Qs @ Qs.T - np.eye(len(idx))).max():.2e}   ascending: {bool(np.all(np.diff(lam) >= 0))}")
    print(line, flush=True)
