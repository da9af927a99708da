"""Compressed sensing MSE against the state-evolution prediction, in the spirit of the reference's
examples/glm experiments (one EP run per seed there; one ENSEMBLE of instances per alpha here).
Usage: python tools/ep_vs_se_curve.py [N] [instances] [rho] [noise_var]
"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import tramp_amd as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rho = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
nv = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-6
ctx = T.default_context()
print(f"sparse CS, Gaussian ensemble, N={N}, rho={rho}, noise var={nv}, {B} instances per alpha")
print(f"{'alpha':>6} {'SE mse':>12} {'EP mse (mean)':>14} {'EP mse (std)':>13} {'EP iters':>9} {'s':>6}")
for alpha in (0.3, 0.4, 0.5, 0.55, 0.6, 0.7, 0.8, 1.0):
    M = int(alpha * N)
    rs = np.random.RandomState(int(alpha * 1000))
    W = rs.randn(M, N) / np.sqrt(N)
    X = rs.standard_normal((B, N)) * rs.binomial(1, rho, (B, N))
    Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((B, M))
    t0 = time.perf_counter()
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.LinearChannel(W, ctx=ctx) @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=300, damping=0.1)
    mse = np.atleast_1d(ep.mse("x", X))
    dt = time.perf_counter() - t0
    se_model = T.glm_state_evolution(alpha=alpha, prior_type="gauss_bernoulli", output_type="gaussian",
                                     prior_rho=rho, output_var=nv)
    se = T.StateEvolution(se_model)
    se.iterate(max_iter=3000, damping=0.1, callback=T.EarlyStopping(tol=1e-14))
    v_se = se.get_variables_data()["x"]["v"]
    print(f"{alpha:6.2f} {float(np.mean(v_se)):12.4e} {mse.mean():14.4e} {mse.std():13.2e} {ep.n_iter:9d} {dt:6.1f}", flush=True)
