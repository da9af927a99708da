"""DFTChannel forward + backward message on the own FFT engine and on rocFFT.  python tools/bench_dft.py [f64|f32]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T  # noqa: E402
from tramp_amd.device import Event  # noqa: E402

np_dt = np.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else np.float64
ctx = T.default_context()
for shape, B in [((1 << 20,), 64), ((1000000,), 64), ((1024, 1024), 64), ((4096, 4096), 4), ((4096,), 64), ((64, 64), 64)]:
    n = int(np.prod(shape))
    res = {}
    for eng in ("own", "rocfft"):
        if eng == "own":
            os.environ["TRAMP_DFT_FFT"] = "own"
        else:
            os.environ["TRAMP_DFT_FFT"] = "rocfft"
        ch = T.DFTChannel(shape=shape, dtype=np_dt, ctx=ctx)
        az, ax = ctx.array(np.full(B, 1.3)), ctx.array(np.full(B, 0.7))
        bz, bx = ctx.randn((B, n), np_dt, seed=1), ctx.randn((B, 2 * n), np_dt, seed=2)
        oa, ox, oz = ctx.empty((B,), np.float64), ctx.empty((B, 2 * n), np_dt), ctx.empty((B, n), np_dt)

        def once():
            ch.compute_forward_message(az, bz, ax, bx, out_a=oa, out_b=ox)
            ch.compute_backward_message(az, bz, ax, bx, out_a=oa, out_b=oz)
        once()
        ctx.sync()
        e0 = Event(ctx).record()
        for _ in range(10):
            once()
        e1 = Event(ctx).record()
        ctx.sync()
        res[eng] = e0.elapsed_ms(e1) / 10
        del ch
    print(f"DFT {'x'.join(map(str, shape))} x{B} {np_dt.__name__}: own {res['own']:.3f} ms, rocFFT {res['rocfft']:.3f} ms "
          f"({res['rocfft'] / res['own']:.2f}x) per forward + backward message")
