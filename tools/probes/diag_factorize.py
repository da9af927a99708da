"""Diagnose a factorisation at N (default 100000): which of {eigh, thin factors, mat-vec at batch b} is off.
Pulls single rows of Vt / Uk to the host and checks them against device mat-vecs at batch 1 (GEMV),
4 and 64 (MFMA)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
M = N // 2
dt = T.device.dtype_code(np.float64)

# 1. mat-vec consistency across batch sizes on a plain random matrix of the same shape
A = ctx.randn((M, N), np.float64, seed=3, scale=1 / np.sqrt(N))
x1 = ctx.randn((1, N), np.float64, seed=5)
xh = x1.get()
for B in (1, 2, 4, 8, 64):
    xb = ctx.array(np.repeat(xh, B, axis=0), np.float64)
    out = ctx.empty((B, M), np.float64)
    ctx._call("tramp_matvec_nt", dt, 0, M, N, B, A.ptr, xb.ptr, out.ptr)
    o = out.get()
    if B == 1:
        ref_nt = o[0].copy()
    print(f"NT batch {B}: max dev from batch-1 result {np.abs(o - ref_nt).max():.3e} (|ref| {np.abs(ref_nt).max():.3e}) "
          f"rows agree {np.abs(o - o[0]).max():.3e}", flush=True)
y1 = ctx.randn((1, M), np.float64, seed=6)
yh = y1.get()
for B in (1, 2, 4, 8, 64):
    yb = ctx.array(np.repeat(yh, B, axis=0), np.float64)
    out = ctx.empty((B, N), np.float64)
    ctx._call("tramp_matvec_nn", dt, 0, M, N, B, A.ptr, yb.ptr, out.ptr)
    o = out.get()
    if B == 1:
        ref_nn = o[0].copy()
    print(f"NN batch {B}: max dev from batch-1 result {np.abs(o - ref_nn).max():.3e} (|ref| {np.abs(ref_nn).max():.3e})", flush=True)
# host check of a few entries of the batch-1 results
rows = [0, 1, M // 2, M - 1]
for r in rows:
    ar = A.row(r).get()
    print(f"NT row {r}: device {ref_nt[r]:.12e} host {ar @ xh[0]:.12e}")
del A

# 2. eigh alone: G = W W^T of a smaller W so that it is cheap, n = M
W = ctx.randn((M, N), np.float64, seed=1, scale=1 / np.sqrt(N))
t0 = time.perf_counter()
lin = T.LinearChannel(W, ctx=ctx, svd="device", keep_W=True)
ctx.sync()
print(f"factorize {time.perf_counter() - t0:.1f}s rank {lin.rank}", flush=True)
s = lin._sing.get()
idx = [0, 1, 2, M // 4, M // 2 - 1, M // 2, M // 2 + 1, 3 * M // 4, M - 2, M - 1]
V = np.stack([lin._Vt.row(j).get() for j in idx])
print("Vt rows norms", np.round(np.linalg.norm(V, axis=1), 12))
print("Vt rows gram offdiag max", np.abs(V @ V.T - np.diag(np.diag(V @ V.T))).max())
# W v_j = s_j u_j : batch-1 GEMV
Uk = lin._Uk
for n_, j in enumerate(idx):
    vj = ctx.array(V[n_][None, :], np.float64)
    out = ctx.empty((1, M), np.float64)
    ctx._call("tramp_matvec_nt", dt, 1, M, N, 1, W.ptr, vj.ptr, out.ptr)
    wv = out.get()[0]
    # column j of Uk (row-major [M, k]) via a strided host read of a few entries is expensive: use u = W v / s
    u = wv / s[j]
    # and check u against Uk^T: t = Uk^T u should be e_j -> use NN with mat = Uk [M,k]
    ub = ctx.array(u[None, :], np.float64)
    e = ctx.empty((1, M), np.float64)
    ctx._call("tramp_matvec_nn", dt, 1, M, M, 1, Uk.ptr, ub.ptr, e.ptr)
    eh = e.get()[0]
    print(f"j={j}: |W v_j| / s_j = {np.linalg.norm(wv) / s[j]:.12f}  (Uk^T u)[j] = {eh[j]:.12f}  max other {np.abs(np.delete(eh, j)).max():.3e}", flush=True)
