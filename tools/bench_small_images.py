"""Small images (launch-bound): a 2-D deconvolution EP iteration eager and as a hipGraph replay, own FFT engine vs rocFFT.
python tools/bench_small_images.py [side] [instances]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = T.default_context()
shape, rho, nv = (side, side), 0.1, 1e-3
n = side * side
rs = np.random.RandomState(3)
f = rs.standard_normal(shape) / np.sqrt(n)
X = rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n))
for eng in ("own", "rocfft"):
    if eng == "rocfft":
        os.environ["TRAMP_CONV_FFT"] = "rocfft"
    else:
        os.environ["TRAMP_CONV_FFT"] = "own"
    ch = T.ConvChannel(f, ctx=ctx)
    Y = ch.convolve(X.reshape((B,) + shape)).reshape(B, n) + np.sqrt(nv) * np.random.RandomState(4).standard_normal((B, n))
    for graph in (False, True):
        model = (T.GaussBernoulliPrior(size=shape, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ ch @ T.SISOVariable(id="z")
                 @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx)
        ep.iterate(max_iter=6, damping=0.1, callback=T.PassCallback(), graph=graph, check_every=1 << 30)
        ctx.sync()
        K = 200
        t0 = time.perf_counter()
        ep.iterate(max_iter=K, damping=0.1, callback=T.PassCallback(), graph=graph, warm_start=True, check_every=1 << 30)
        ctx.sync()
        dt = time.perf_counter() - t0
        print(f"{side}x{side} x{B} {eng:6s} graph={graph}: {dt / K * 1e6:8.1f} us per EP iteration, mse {float(np.mean(ep.mse('x', X))):.3e}")
