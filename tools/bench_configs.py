"""The other BASELINE.json configurations at scale on one MI355X (context for DESIGN.md; the
headline bench line is bench.py):
  cs1bit   : 1-bit compressed sensing (GaussBernoulli + LinearChannel + SgnLikelihood), alpha=0.7, fp32
  twolayer : Gaussian prior -> LinearChannel(W1) -> ReluChannel -> LinearChannel(W2) -> Gaussian likelihood
Usage: python tools/bench_configs.py cs1bit|twolayer [N] [instances] [dtype]
"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T

which = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
np_dt = np.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else np.float32
ctx = T.default_context()
dt = T.device.dtype_code(np_dt)
rs = np.random.RandomState(7)
t0 = time.perf_counter()


def dev_matvec(Wd, M_, N_, Xh):
    Xd = ctx.array(Xh.astype(np_dt), np_dt)
    Zd = ctx.empty((Xh.shape[0], M_), np_dt)
    ctx._call("tramp_matvec_nt", dt, 0, M_, N_, Xh.shape[0], Wd.ptr, Xd.ptr, Zd.ptr)
    return Zd.get().astype(np.float64)


if which == "cs1bit":
    M = int(0.7 * N) // 4 * 4
    W = ctx.randn((M, N), np_dt, seed=11, scale=N ** -0.5)
    X = rs.standard_normal((B, N)) * rs.binomial(1, 0.25, (B, N))
    Y = np.sign(dev_matvec(W, M, N, X))
    lin = T.LinearChannel(W, dtype=np_dt, ctx=ctx, svd="device"); del W
    model = (T.GaussBernoulliPrior(size=N, rho=0.25, dtype=np_dt, ctx=ctx) @ T.SISOVariable(id="x") @ lin
             @ T.SISOVariable(id="z") @ T.SgnLikelihood(y=Y.astype(np_dt), dtype=np_dt, ctx=ctx)).to_model()
    desc = f"1-bit CS N={N} M={M}"
    def score(ep):      # sign retrieval: overlap of the normalised estimate
        r = ep.get_variables_data()["x"]["r"].reshape(B, N)
        return float(np.mean(np.sum(r * X, 1) / (np.linalg.norm(r, axis=1) * np.linalg.norm(X, axis=1))))
elif which == "twolayer":
    M1, M2 = N // 2, N // 4
    W1 = ctx.randn((M1, N), np_dt, seed=12, scale=N ** -0.5)
    W2 = ctx.randn((M2, M1), np_dt, seed=13, scale=M1 ** -0.5)
    X = rs.standard_normal((B, N))
    A1 = np.maximum(0, dev_matvec(W1, M1, N, X))
    Y = dev_matvec(W2, M2, M1, A1) + 0.1 * rs.standard_normal((B, M2))
    l1 = T.LinearChannel(W1, dtype=np_dt, ctx=ctx, svd="device"); del W1
    l2 = T.LinearChannel(W2, dtype=np_dt, ctx=ctx, svd="device"); del W2
    model = T.MultiLayerModel([T.GaussianPrior(size=N, dtype=np_dt, ctx=ctx), l1, T.ReluChannel(dtype=np_dt, ctx=ctx),
                               l2, T.GaussianLikelihood(y=Y.astype(np_dt), var=1e-2, dtype=np_dt, ctx=ctx)],
                              ids=["x", "z1", "a1", "z2", "y"])
    desc = f"two-layer relu N={N} M1={M1} M2={M2}"
    def score(ep):
        return float(np.mean(ep.mse("a1", A1)))
else:
    raise SystemExit(__doc__)
ctx.sync()
setup = time.perf_counter() - t0
ep = T.ExpectationPropagation(model, ctx=ctx)
ep.iterate(max_iter=3, callback=T.PassCallback(), damping=0.1, check_every=1 << 30)
ctx._call("tramp_profile_enable", 1)
ctx.sync(); t0 = time.perf_counter()
K = 20
ep.iterate(max_iter=K, callback=T.PassCallback(), damping=0.1, warm_start=True, check_every=1 << 30)
ctx.sync(); dt_s = time.perf_counter() - t0
import ctypes as C
cnt, ms, by, fl = (C.c_int * 8)(), (C.c_double * 8)(), (C.c_double * 8)(), (C.c_double * 8)()
ctx._call("tramp_profile_report", cnt, ms, by, fl)
tags = ["ew", "matvec_nt", "matvec_nn", "lin_scalars", "var_post", "other"]
print(f"{desc} {np_dt.__name__} B={B}: {dt_s/K*1e3:.3f} ms/iteration, {N*B*K/dt_s:.3e} msg-components/s, setup {setup:.1f} s, "
      f"score {score(ep):.4f}")
print("   per-tag ms:", {tags[i]: round(ms[i], 2) for i in range(6) if cnt[i]},
      "| ew GB/s alg:", round(by[0] / max(ms[0], 1e-9) / 1e6), "| matvec TFLOP/s:",
      round((fl[1] + fl[2]) / max(ms[1] + ms[2], 1e-9) / 1e9, 1))
