"""The headline workload run to convergence: 64-instance sparse GLM (rho = 0.05, alpha = 0.5, noise 1e-2,
damping 0.1) at N = 65536 with the default stopping rule; ensemble MSE against the state-evolution fixed point.
Usage: python tools/headline_convergence.py [N] [instances]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import tramp_amd as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M, rho, nv = N // 2, 0.05, 1e-2
ctx = T.default_context()
rs = np.random.RandomState(1)
W = ctx.randn((M, N), np.float64, seed=1234, scale=1.0 / np.sqrt(N))
X = rs.standard_normal((B, N)) * rs.binomial(1, rho, (B, N))
Xd, Zd = ctx.array(X, np.float64), ctx.empty((B, M), np.float64)     # keep Xd alive until the launch has run
ctx._call("tramp_matvec_nt", T.device.dtype_code(np.float64), 0, M, N, B, W.ptr, Xd.ptr, Zd.ptr)
Y = Zd.get() + np.sqrt(nv) * rs.standard_normal((B, M))
t0 = time.perf_counter()
lin = T.LinearChannel(W, ctx=ctx, svd="device")
del W
ctx.sync()
t_fact = time.perf_counter() - t0
model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ lin @ T.SISOVariable(id="z")
         @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
ep = T.ExpectationPropagation(model, ctx=ctx)
t0 = time.perf_counter()
ep.iterate(max_iter=1000, damping=0.1)                 # EarlyStoppingEP(tol=1e-6), per-instance stopping
ctx.sync()
t_ep = time.perf_counter() - t0
mse = np.asarray(ep.mse("x", X))
se = T.StateEvolution(T.glm_state_evolution(alpha=0.5, prior_type="gauss_bernoulli", output_type="gaussian",
                                            prior_rho=rho, output_var=nv)).iterate(max_iter=1000)
v_se = se.get_variables_data()["x"]["v"]
print(f"N={N} instances={B}: factorisation {t_fact:.1f} s; EP converged in {ep.n_iter} iterations "
      f"(per instance {ep.n_iter_instance.min()}..{ep.n_iter_instance.max()}), {t_ep:.2f} s = {t_ep/ep.n_iter*1e3:.2f} ms/iteration")
print(f"  ensemble MSE {mse.mean():.6e} (min {mse.min():.4e}, max {mse.max():.4e}); EP's own v {np.mean(ep.get_variables_data(want_r=False)['x']['v']):.6e}; "
      f"state-evolution fixed point {v_se:.6e} ({se.n_iter} iterations)")
