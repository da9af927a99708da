"""Sparse deconvolution at config-2 scale (n = 10^6) through the FFT-backed ConvChannel:
GaussBernoulli prior -> ConvChannel(filter) -> Gaussian likelihood, an ensemble of instances.
Usage: python tools/bench_conv.py [n] [instances] [f64|f32] [filter_taps] [gaussian|sgn]
Prints ms / iteration, message components / s, the per-tag kernel time and the effective
HBM rate against the algorithmic bytes of the iteration.
"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import tramp_amd as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
np_dt = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else np.float64
taps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
output = sys.argv[5] if len(sys.argv) > 5 else "gaussian"     # sgn: 1-bit measurements (BASELINE configs[2] with a circulant operator)
rho, nv = 0.05, 1e-3
ctx = T.default_context()
rs = np.random.RandomState(3)
f = np.zeros(n)
if taps:
    f[:taps] = rs.randn(taps) / np.sqrt(taps)          # short random blur
else:
    f[:] = rs.randn(n) / np.sqrt(n)                    # dense random circulant (a structured sensing matrix)
t0 = time.perf_counter()
ch = T.ConvChannel(f, dtype=np_dt, ctx=ctx)
X = (rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n))).astype(np_dt)
Xd = ctx.array(X, np_dt)
Yd = ch.convolve(Xd)
if output == "sgn":
    Y = np.sign(Yd.get()).astype(np_dt)
    lik = T.SgnLikelihood(y=Y, dtype=np_dt, ctx=ctx)
else:
    Y = Yd.get() + np.sqrt(nv) * rs.standard_normal((B, n)).astype(np_dt)
    lik = T.GaussianLikelihood(y=Y, var=nv, dtype=np_dt, ctx=ctx)
model = (T.GaussBernoulliPrior(size=n, rho=rho, dtype=np_dt, ctx=ctx) @ T.SISOVariable(id="x") @ ch
         @ T.SISOVariable(id="z") @ lik).to_model()
ep = T.ExpectationPropagation(model, ctx=ctx)
ctx.sync()
setup = time.perf_counter() - t0
never = 1 << 30
for graph in (False, True):
    ep.iterate(max_iter=4, callback=T.PassCallback(), check_every=never, graph=graph)
    ctx.sync()
    K = 20
    t0 = time.perf_counter()
    ep.iterate(max_iter=K, callback=T.PassCallback(), warm_start=True, check_every=never, graph=graph)
    ctx.sync()
    dt_s = time.perf_counter() - t0
    if output == "sgn":       # sign measurements lose the scale: report 1 - overlap of the normalised estimate
        r = np.asarray(ep.get_variables_data()["x"]["r"]).reshape(B, n)
        mse = float(1 - np.mean(np.sum(r * X, 1) / (np.linalg.norm(r, axis=1) * np.linalg.norm(X, axis=1) + 1e-300)))
    else:
        mse = float(np.mean(ep.mse("x", X)))
    print(f"conv EP n={n} B={B} {np_dt.__name__} taps={taps or n} output={output} graph={graph}: {dt_s/K*1e3:.3f} ms/iteration, "
          f"{n*B*K/dt_s:.3e} msg-components/s, mse {mse:.3e}, setup {setup:.1f} s")
# per-tag timing (eager, HIP events)
ctx._call("tramp_profile_enable", 1)
ep.iterate(max_iter=10, callback=T.PassCallback(), warm_start=True, check_every=never)
ctx.sync()
NT = 8
cnt, ms, by, fl = (C.c_int * NT)(), (C.c_double * NT)(), (C.c_double * NT)(), (C.c_double * NT)()
ctx._call("tramp_profile_report", cnt, ms, by, fl)
ctx._call("tramp_profile_enable", 0)
tags = ["ew", "matvec_nt", "matvec_nn", "lin_scalars", "var_post", "other", "fft"]
tot_ms = sum(ms[i] for i in range(7))
tot_by = sum(by[i] for i in range(7))
print("   per-tag (10 iterations):", {tags[i]: dict(n=cnt[i], ms=round(ms[i], 3), GBps=round(by[i] / max(ms[i], 1e-9) / 1e6))
                                      for i in range(7) if cnt[i]})
print(f"   tagged kernel time {tot_ms/10:.3f} ms/iteration, algorithmic {tot_by/10/1e9:.3f} GB/iteration "
      f"-> {tot_by/max(tot_ms,1e-9)/1e6:.0f} GB/s effective")
