"""Launch the hot-path kernels a few times on synthetic operands (no library calls), for the
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE crash inside rocSOLVER under the profiler,
so the counter passes avoid the factorisation)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = N // 2
ctx = T.default_context()
dt = T.device.dtype_code(np.float64)
Vt = ctx.randn((M, N), seed=1, scale=N ** -0.5)
x = ctx.randn((B, N), seed=2)
y = ctx.randn((B, M), seed=3)
t = ctx.empty((B, M)); r = ctx.empty((B, N))
for _ in range(3):
    ctx._call("tramp_matvec_nt", dt, 0, M, N, B, Vt.ptr, x.ptr, t.ptr)       # t = Vt x   (MFMA NT)
    ctx._call("tramp_matvec_nn", dt, 0, M, N, B, Vt.ptr, y.ptr, r.ptr)       # r = Vt^T y (MFMA NN)
x1 = ctx.randn((1, N), seed=4); y1 = ctx.randn((1, M), seed=5)
t1 = ctx.empty((1, M)); r1 = ctx.empty((1, N))
for _ in range(3):
    ctx._call("tramp_matvec_nt", dt, 0, M, N, 1, Vt.ptr, x1.ptr, t1.ptr)     # GEMV
    ctx._call("tramp_matvec_nn", dt, 0, M, N, 1, Vt.ptr, y1.ptr, r1.ptr)
NE = 1_000_000
for Bd in (1, 64):
    bx = ctx.randn((Bd, NE), seed=6, scale=1.5); ax = ctx.scalars(1.3, Bd)
    bz = ctx.randn((Bd, NE), seed=7, scale=1.5)
    a_new = ctx.empty((Bd,)); b_new = ctx.empty((Bd, NE))
    p = T.GaussBernoulliPrior(size=NE, rho=0.05, ctx=ctx)
    relu = T.ReluChannel(ctx=ctx)
    for _ in range(3):
        p.compute_forward_message(ax, bx, out_a=a_new, out_b=b_new)
        relu.compute_backward_message(ax, bz, ax, bx, out_a=a_new, out_b=b_new)
ctx.sync()
print("done")
