"""The two f64 dense passes of the headline shape, three launches each on Gaussian operands and then on all-zero operands
(for a rocprofv3 --pmc pass: the kernel trace orders the dispatches): python tools/pmc_matvec_power.py"""
import sys

import numpy as np

sys.path.insert(0, '.')
import tramp_amd as T

N, B = 100000, 64
M = N // 2
ctx = T.default_context()
dt = T.device.dtype_code(np.float64)
for what in ("random", "zero"):
    if what == "random":
        Vt, x, y = ctx.randn((M, N), np.float64, seed=1, scale=N ** -0.5), ctx.randn((B, N), np.float64, seed=2), ctx.randn((B, M), np.float64, seed=3)
    else:
        Vt, x, y = ctx.zeros((M, N), np.float64), ctx.zeros((B, N), np.float64), ctx.zeros((B, M), np.float64)
    t, r = ctx.empty((B, M), np.float64), ctx.empty((B, N), np.float64)
    for _ in range(3):
        ctx._call("tramp_matvec_nt", dt, 2, M, N, B, Vt.ptr, x.ptr, t.ptr)
    for _ in range(3):
        ctx._call("tramp_matvec_nn", dt, 2, M, N, B, Vt.ptr, y.ptr, r.ptr)
    ctx.sync()
    del Vt, x, y, t, r
print("done")
