"""Sustained time of the headline's pass pair (row-dot then column pass over Vt, f64, 64 instances), for A/B runs of two
library builds in separate processes: TRAMP_HIP_LIB=... python tools/ab_matvec_pair.py [pairs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import tramp_amd as T

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 150
N, B = 100000, 64
M = N // 2
ctx = T.default_context()
dt = T.device.dtype_code(np.float64)
Vt = ctx.randn((M, N), np.float64, seed=1, scale=N ** -0.5)
x, y = ctx.randn((B, N), np.float64, seed=2), ctx.randn((B, M), np.float64, seed=3)
t, r = ctx.empty((B, M), np.float64), ctx.empty((B, N), np.float64)
for _ in range(5):
    ctx._call("tramp_matvec_nt", dt, 2, M, N, B, Vt.ptr, x.ptr, t.ptr)
    ctx._call("tramp_matvec_nn", dt, 2, M, N, B, Vt.ptr, y.ptr, r.ptr)
ctx.sync()
t0 = time.perf_counter()
for _ in range(pairs):
    ctx._call("tramp_matvec_nt", dt, 2, M, N, B, Vt.ptr, x.ptr, t.ptr)
    ctx._call("tramp_matvec_nn", dt, 2, M, N, B, Vt.ptr, y.ptr, r.ptr)
ctx.sync()
el = time.perf_counter() - t0
print(f"{os.environ.get('TRAMP_HIP_LIB', 'default library'):40s} {pairs} pairs: {el / pairs * 1e3:7.3f} ms per pair  "
      f"({4.0 * M * N * B * pairs / el / 1e12:5.1f} TFLOP/s)  checksum {float(t.get()[3, 7]):.12e} {float(r.get()[5, 11]):.12e}", flush=True)
