"""The two dense passes of the headline shape on the int8-digit engine, a few launches each (for rocprofv3 passes):
python tools/pmc_digits.py [N] [B] [digits]"""
import sys

import numpy as np

sys.path.insert(0, '.')
import tramp_amd as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
D = int(sys.argv[3]) if len(sys.argv) > 3 else 7
M = N // 2
ctx = T.default_context()
dt = T.device.dtype_code(np.float64)
Vt = ctx.randn((M, N), np.float64, seed=1, scale=N ** -0.5)
x, y = ctx.randn((B, N), np.float64, seed=2), ctx.randn((B, M), np.float64, seed=3)
t, r = ctx.empty((B, M), np.float64), ctx.empty((B, N), np.float64)
ctx.attach_digits(Vt, D, 3)
for _ in range(4):
    ctx._call("tramp_matvec_nt", dt, 3, M, N, B, Vt.ptr, x.ptr, t.ptr)
    ctx._call("tramp_matvec_nn", dt, 3, M, N, B, Vt.ptr, y.ptr, r.ptr)
ctx.sync()
print("done", ctx.sliced_runs())
