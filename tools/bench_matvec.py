"""Time the matrix primitives in isolation: python tools/bench_matvec.py N B [dtype]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
N = int(sys.argv[1]); B = int(sys.argv[2]); M = N // 2
np_dt = np.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else np.float64
ALGO = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = T.default_context()
dt = T.device.dtype_code(np_dt)
es = np.dtype(np_dt).itemsize
Vt = ctx.randn((M, N), np_dt, seed=1, scale=N ** -0.5)
U = ctx.randn((M, M), np_dt, seed=8, scale=N ** -0.5)
x = ctx.randn((B, N), np_dt, seed=2); y = ctx.randn((B, M), np_dt, seed=3)
t = ctx.empty((B, M), np_dt); r = ctx.empty((B, N), np_dt)
def timeit(name, fn, R, C, reps=5):
    fn(); fn(); ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(reps): fn()
    e1.record()
    ms = e0.elapsed_ms(e1) / reps
    print(f"{name:10s} algo={ALGO} {np_dt.__name__} R={R} C={C} B={B}: {ms:8.3f} ms  {2.0*R*C*B/ms/1e9:7.2f} TFLOP/s  {R*C*es/ms/1e6:7.0f} GB/s(mat)", flush=True)
timeit("NT(Vt)", lambda: ctx._call("tramp_matvec_nt", dt, ALGO, M, N, B, Vt.ptr, x.ptr, t.ptr), M, N)
timeit("NN(Vt)", lambda: ctx._call("tramp_matvec_nn", dt, ALGO, M, N, B, Vt.ptr, y.ptr, r.ptr), M, N)
timeit("NT(U)", lambda: ctx._call("tramp_matvec_nt", dt, ALGO, M, M, B, U.ptr, y.ptr, t.ptr), M, M)
timeit("NN(U)", lambda: ctx._call("tramp_matvec_nn", dt, ALGO, M, M, B, U.ptr, y.ptr, t.ptr), M, M)
# the row-dot pass t = Vt x as a column pass over the transposed copy V [N, M] (same flops, same bytes)
V = ctx.randn((N, M), np_dt, seed=9, scale=N ** -0.5)
timeit("NN(V=Vt^T)", lambda: ctx._call("tramp_matvec_nn", dt, ALGO, N, M, B, V.ptr, x.ptr, t.ptr), N, M)
