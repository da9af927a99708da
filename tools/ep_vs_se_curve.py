"""Compressed sensing MSE against the state-evolution prediction, in the spirit of the reference's
examples/glm experiments (one EP run per seed there; one ENSEMBLE of instances per alpha here).
Usage: python tools/ep_vs_se_curve.py [N] [instances] [rho] [noise_var] [gaussian|sgn]
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
output = sys.argv[5] if len(sys.argv) > 5 else "gaussian"
ctx = T.default_context()
print(f"sparse CS, Gaussian ensemble, N={N}, rho={rho}, output={output}" + (f", noise var={nv}" if output == "gaussian" else "")
      + f", {B} instances per alpha")
print(f"{'alpha':>6} {'SE mse':>12} {'EP mse (mean)':>14} {'EP mse (std)':>13} {'EP iters':>9} {'s':>6}")
for alpha in ((0.3, 0.4, 0.5, 0.55, 0.6, 0.7, 0.8, 1.0) if output == "gaussian" else (0.5, 1.0, 2.0, 3.0)):
    M = int(alpha * N)
    rs = np.random.RandomState(int(alpha * 1000))
    W = rs.randn(M, N) / np.sqrt(N)
    X = rs.standard_normal((B, N)) * rs.binomial(1, rho, (B, N))
    Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((B, M)) if output == "gaussian" else np.sign(X @ W.T)
    lik = T.GaussianLikelihood(y=Y, var=nv, ctx=ctx) if output == "gaussian" else T.SgnLikelihood(y=Y, ctx=ctx)
    t0 = time.perf_counter()
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.LinearChannel(W, ctx=ctx) @ T.SISOVariable(id="z")
             @ lik).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=300, damping=0.1)
    mse = np.atleast_1d(ep.mse("x", X))
    dt = time.perf_counter() - t0
    okw = dict(output_var=nv) if output == "gaussian" else {}
    se_model = T.glm_state_evolution(alpha=alpha, prior_type="gauss_bernoulli", output_type=output, prior_rho=rho, **okw)
    se = T.StateEvolution(se_model)
    # (the sign likelihood's SE needs az > 1/tau_z from the first sweep, as in the reference: start from a
    #  large a and do not damp)
    if output == "gaussian":
        se.iterate(max_iter=3000, damping=0.1, callback=T.EarlyStopping(tol=1e-14))
    else:
        se.iterate(max_iter=3000, callback=T.EarlyStopping(tol=1e-14), initializer=T.ConstantInit(a=10.0))
    v_se = se.get_variables_data()["x"]["v"]
    print(f"{alpha:6.2f} {float(np.mean(v_se)):12.4e} {mse.mean():14.4e} {mse.std():13.2e} {ep.n_iter:9d} {dt:6.1f}", flush=True)
