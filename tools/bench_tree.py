"""Tree-structured EP at image scale: piecewise-constant images observed in Gaussian noise, recovered with a
Gauss-Bernoulli ("sgrad") or group-L21 ("tv") prior on the gradient field - the reference's sparse-gradient /
total-variation models (examples/sparse/plot_sparse_gradient.py, models/total_variation_model.py).
Usage: python tools/bench_tree.py [side] [instances] [f64|f32] [sgrad|tv]
Prints ms / iteration, message components / s (pixels x instances x iterations / s), the per-tag kernel time.
"""
import ctypes as C
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
np_dt = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else np.float64
kind = sys.argv[4] if len(sys.argv) > 4 else "sgrad"
shape, nv, rho = (side, side), 1e-2, 0.02
n = side * side
ctx = T.default_context()
rs = np.random.RandomState(5)
blocks = 16
X = np.kron(rs.standard_normal((B, blocks, blocks)), np.ones((side // blocks, side // blocks)))
Y = X + np.sqrt(nv) * rs.standard_normal(X.shape)
kw = dict(dtype=np_dt, ctx=ctx)
t0 = time.perf_counter()
zprior = T.GaussBernoulliPrior(size=(2,) + shape, rho=rho, **kw) if kind == "sgrad" else \
    T.MAP_L21NormPrior(size=(2,) + shape, gamma=0.5, **kw)
model = (T.GaussianPrior(size=shape, **kw) @ T.SIMOVariable(id="x", n_next=2) @ (
    T.GaussianLikelihood(y=Y.reshape(B, n).astype(np_dt), var=nv, **kw) + (
        T.GradientChannel(shape=shape, **kw) + zprior) @ T.MILeafVariable(id="z", n_prev=2))).to_model()
ep = T.ExpectationPropagation(model, ctx=ctx)
ctx.sync()
setup = time.perf_counter() - t0
never = 1 << 30
TAGS = ["ew", "matvec_nt", "matvec_nn", "lin_scalars", "var_post", "factorize", "fft"]
init = T.ConstantInit(a=0.1, b=1.0)
for graph in (False, True):
    ep.iterate(max_iter=4, callback=T.PassCallback(), check_every=never, graph=graph, damping=0.3, initializer=init)
    ctx.sync()
    K = 20
    if not graph:
        ctx._call("tramp_profile_enable", 1)
    t0 = time.perf_counter()
    ep.iterate(max_iter=K, callback=T.PassCallback(), warm_start=True, check_every=never, graph=graph, damping=0.3)
    ctx.sync()
    dt_s = time.perf_counter() - t0
    NT = len(TAGS)
    cnt, ms, by, fl = (C.c_int * NT)(), (C.c_double * NT)(), (C.c_double * NT)(), (C.c_double * NT)()
    if not graph:
        ctx._call("tramp_profile_report", cnt, ms, by, fl)
        ctx._call("tramp_profile_enable", 0)
    mse = float(np.mean(ep.mse("x", X.reshape(B, n))))
    print(f"tree EP ({kind}) {side}x{side} B={B} {np_dt.__name__} graph={graph}: {dt_s/K*1e3:.3f} ms/iteration, "
          f"{n*B*K/dt_s:.3e} msg-components/s, mse {mse:.3e} (observation {nv:g}), setup {setup:.1f} s")
    if not graph:
        print("   per-tag ms/iteration:", {TAGS[i]: round(ms[i] / K, 3) for i in range(NT) if cnt[i]})
