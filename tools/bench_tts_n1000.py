"""Where the time of the N = 1000 time-to-solution run goes: python tools/bench_tts_n1000.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
N, M, rho, nv = 1000, 500, 0.05, 1e-2
for restore in ((None, True, False) if len(sys.argv) > 1 else (False, True, None)):
    for graph in (False, True):
        ts = []
        for seed in range(6):
            rs = np.random.RandomState(seed)
            W = rs.randn(M, N) / np.sqrt(N)
            x = rs.standard_normal(N) * rs.binomial(1, rho, N)
            y = W @ x + np.sqrt(nv) * rs.standard_normal(M)
            ctx.sync(); t0 = time.perf_counter()
            lin = T.LinearChannel(W, ctx=ctx, svd="device")
            ctx.sync(); t1 = time.perf_counter()
            model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ lin
                     @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=y, var=nv, ctx=ctx)).to_model()
            ep = T.ExpectationPropagation(model, ctx=ctx)
            t2 = time.perf_counter()
            ep.iterate(max_iter=1000, callback=T.EarlyStoppingEP(tol=1e-6), damping=0.1, check_every=4, graph=graph,
                       restore_messages=restore)
            ctx.sync(); t3 = time.perf_counter()
            r = ep.get_variables_data(["x"])["x"]["r"]
            t4 = time.perf_counter()
            if seed:
                ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, ep.n_iter))
            del ep, model, lin
        a = np.median(np.array(ts), axis=0)
        print(f"restore={restore} graph={graph}: factorise {a[0]*1e3:.1f} ms, build {a[1]*1e3:.2f} ms, iterate {a[2]*1e3:.2f} ms "
              f"({a[4]:.0f} it), read {a[3]*1e3:.2f} ms", flush=True)
