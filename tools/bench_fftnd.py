"""The 2-D / 3-D channels on the own FFT engine against the same channels on rocFFT: one forward + one backward message
of a GradientChannel / ConvChannel per shape (what an EP iteration of a total-variation or deconvolution model spends in
the channel).  python tools/bench_fftnd.py [f64|f32] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T  # noqa: E402
from tramp_amd.device import Event  # noqa: E402

np_dt = np.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else np.float64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = T.default_context()
CASES = [("grad", (4096, 4096), 4), ("grad", (1024, 1024), 64), ("grad", (256, 256), 64), ("conv", (4096, 4096), 4),
         ("conv", (1024, 1024), 64), ("conv", (128, 128, 128), 8), ("grad", (128, 128, 128), 8), ("conv", (64, 64), 64)]


def run(kind, shape, B, engine):
    if engine == "rocfft":
        os.environ["TRAMP_CONV_FFT"] = "rocfft"
    else:
        os.environ["TRAMP_CONV_FFT"] = "own"
    n, d = int(np.prod(shape)), len(shape)
    if kind == "grad":
        ch = T.GradientChannel(shape, dtype=np_dt, ctx=ctx)
        nx = d * n
    else:
        f = np.zeros(shape)
        f[tuple(slice(0, 3) for _ in shape)] = np.random.RandomState(0).standard_normal((3,) * d)
        ch = T.ConvChannel(f, dtype=np_dt, ctx=ctx)
        nx = n
    az, ax = ctx.array(np.full(B, 1.3)), ctx.array(np.full(B, 0.7))
    bz, bx = ctx.randn((B, n), np_dt, seed=1), ctx.randn((B, nx), np_dt, seed=2)
    oa, ob_x, ob_z = ctx.empty((B,), np.float64), ctx.empty((B, nx), np_dt), ctx.empty((B, n), np_dt)
    def once():
        bz.version = getattr(bz, "version", 0) + 1            # no projection reuse: every transform runs
        bx.version = getattr(bx, "version", 0) + 1
        ch.compute_forward_message(az, bz, ax, bx, out_a=oa, out_b=ob_x)
        bz.version += 1
        bx.version += 1
        ch.compute_backward_message(az, bz, ax, bx, out_a=oa, out_b=ob_z)
    once()
    ctx.sync()
    e0, e1 = Event(ctx).record(), None
    for _ in range(reps):
        once()
    e1 = Event(ctx).record()
    ctx.sync()
    ms = e0.elapsed_ms(e1) / reps
    out = ob_z.get()[0, :8].copy()
    del ch
    return ms, out


if len(sys.argv) > 3 and sys.argv[3] == "one":
    # profiling: one case on the engine TRAMP_CONV_FFT selects
    # python tools/bench_fftnd.py f64 10 one [conv|grad] [AxBxC] [instances]
    eng = "rocfft" if os.environ.get("TRAMP_CONV_FFT") == "rocfft" else "own"
    os.environ.setdefault("TRAMP_CONV_FFT", "own")
    kind1 = sys.argv[4] if len(sys.argv) > 4 else "grad"
    shape1 = tuple(int(v) for v in sys.argv[5].split("x")) if len(sys.argv) > 5 else (4096, 4096)
    B1 = int(sys.argv[6]) if len(sys.argv) > 6 else 4
    print(kind1, "x".join(map(str, shape1)), f"x{B1}", np_dt.__name__, eng, run(kind1, shape1, B1, eng)[0], "ms")
    sys.exit(0)

for kind, shape, B in CASES:
    try:
        t_own, o1 = run(kind, shape, B, "own")
        t_roc, o2 = run(kind, shape, B, "rocfft")
    except Exception as e:                                   # noqa: BLE001
        print(kind, shape, B, "failed:", e)
        continue
    n = int(np.prod(shape))
    nsig = B * (2 * (1 + (len(shape) if kind == "grad" else 1)) + (len(shape) if kind == "grad" else 1) + 1)
    esz = np.dtype(np_dt).itemsize
    print(f"{kind} {'x'.join(map(str, shape))} x{B} {np_dt.__name__}: own {t_own:.3f} ms, rocFFT {t_roc:.3f} ms "
          f"({t_roc / t_own:.2f}x); {nsig} transforms of {n} points per fwd+bwd pair -> "
          f"{nsig * n * esz * 2 / t_own / 1e6:.0f} GB/s if each moved its signal once in, once out; "
          f"max |own - rocFFT| on 8 entries {np.abs(o1 - o2).max():.2e}")
