"""The two dense passes on the f64 matrix cores and on the int8-digit copy, by ensemble size:
python tools/bench_matvec_digits.py [N] [digits]"""
import sys

import numpy as np

sys.path.insert(0, '.')
import tramp_amd as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 7
M = N // 2
ctx = T.default_context()
dt = T.device.dtype_code(np.float64)
Vt = ctx.randn((M, N), np.float64, seed=1, scale=N ** -0.5)
ctx.attach_digits(Vt, D, 3)


def timeit(fn, reps=6):
    fn(); fn(); ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    return e0.elapsed_ms(e1) / reps


for B in (1, 2, 3, 8, 16, 24, 32, 40, 48, 64, 128, 512):
    x, y = ctx.randn((B, N), np.float64, seed=2), ctx.randn((B, M), np.float64, seed=3)
    t, r = ctx.empty((B, M), np.float64), ctx.empty((B, N), np.float64)
    row = f"B={B:4d}"
    for algo, name in ((1 if B <= 2 else 2, "f64 GEMV" if B <= 2 else "f64 MFMA"), (3, f"{D} digits")):
        a = timeit(lambda: ctx._call("tramp_matvec_nt", dt, algo, M, N, B, Vt.ptr, x.ptr, t.ptr))
        b = timeit(lambda: ctx._call("tramp_matvec_nn", dt, algo, M, N, B, Vt.ptr, y.ptr, r.ptr))
        row += f"   {name}: row-dot {a:7.3f} ms  column {b:7.3f} ms"
    print(row, flush=True)
