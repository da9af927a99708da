"""EP iterations/s at the reference's own benchmark size (N=1000, alpha=0.5), eager vs hipGraph."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
for N, B in ((1000, 1), (1000, 64), (4000, 1)):
    M, rho, nv = N // 2, 0.05, 1e-2
    rs = np.random.RandomState(1)
    W = rs.randn(M, N) / np.sqrt(N)
    X = rs.standard_normal((B, N)) * rs.binomial(1, rho, (B, N))
    Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((B, M))
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.LinearChannel(W, ctx=ctx, svd="device") @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    for graph in (False, True):
        ep.iterate(max_iter=5, callback=T.PassCallback(), damping=0.1)
        ctx.sync(); t0 = time.perf_counter()
        K = 500
        ep.iterate(max_iter=K, callback=T.PassCallback(), damping=0.1, check_every=1 << 30, graph=graph)
        ctx.sync(); dt = time.perf_counter() - t0
        print(f"N={N} B={B} graph={graph}: {dt/K*1e6:8.1f} us/iteration  {K/dt:9.0f} it/s  {N*B*K/dt:.3e} msg-components/s", flush=True)
    ep.iterate(max_iter=1000, damping=0.1)
    print(f"   converged in n_iter={ep.n_iter}, mse={np.mean(ep.mse('x', X)):.3e}")
