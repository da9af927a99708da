"""The dense passes on random operands against the same launches on all-zero operands: what the package power costs.
Cycles per MFMA do not depend on the operands (MI355X_MICROARCH.md, 'Matrix cores'), the clock the chip holds does.
python tools/bench_matvec_power.py [N] [B]   ->  profiles/r04/matvec_random_vs_zero_operands.txt"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import tramp_amd as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = N // 2
ctx = T.default_context()
PEAK = {np.float64: 78.6, np.float32: 157.3}


def timeit(fn, reps=10):
    fn(); fn(); ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    return e0.elapsed_ms(e1) / reps


for np_dt in (np.float64, np.float32):
    dt = T.device.dtype_code(np_dt)
    for what in ("random", "zero"):
        if what == "random":
            Vt = ctx.randn((M, N), np_dt, seed=1, scale=N ** -0.5)
            x = ctx.randn((B, N), np_dt, seed=2)
            y = ctx.randn((B, M), np_dt, seed=3)
        else:
            Vt, x, y = ctx.zeros((M, N), np_dt), ctx.zeros((B, N), np_dt), ctx.zeros((B, M), np_dt)
        t, r = ctx.empty((B, M), np_dt), ctx.empty((B, N), np_dt)
        for name, fn in (("row-dot (NT)", lambda: ctx._call("tramp_matvec_nt", dt, 2, M, N, B, Vt.ptr, x.ptr, t.ptr)),
                         ("column  (NN)", lambda: ctx._call("tramp_matvec_nn", dt, 2, M, N, B, Vt.ptr, y.ptr, r.ptr))):
            time.sleep(2.0)                       # let the package cool between measurements
            ms = timeit(fn)
            tf = 2.0 * M * N * B / ms / 1e9
            print(f"{name} {np_dt.__name__} {M} x {N} x {B}  {what:6s} operands: {ms:7.3f} ms  {tf:6.1f} TFLOP/s = {tf / PEAK[np_dt]:.3f} of the "
                  f"matrix peak, matrix streamed at {M * N * np.dtype(np_dt).itemsize / ms / 1e6:5.0f} GB/s", flush=True)
        del Vt, x, y, t, r
