"""Which switch changes the undamped batch-3 EP result?  python tools/probes/diag_fused.py"""
import os, subprocess, sys
code = r'''
import sys, numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
from oracle import tramp_oracle as orc
ctx = T.default_context()
rs = np.random.RandomState(3)
N, M, rho, nv = 4352, 192, 0.05, 1e-2
W = orc.gaussian_ensemble(rs, M, N)
for batch, damping in ((3, None), (3, 0.1), (5, None), (16, None), (64, None)):
    rs2 = np.random.RandomState(3); orc.gaussian_ensemble(rs2, M, N)
    X = np.stack([orc.gb_sample(rs2, N, rho) for _ in range(batch)])
    Y = X @ W.T + np.sqrt(nv) * rs2.standard_normal((batch, M))
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ T.LinearChannel(W, ctx=ctx, svd="host")
             @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    errs = []
    lin = orc.LinearFactor(W)
    for n_it in (1, 2, 3, 12):
        ep.iterate(max_iter=n_it, callback=T.PassCallback(), damping=damping)
        r = np.atleast_2d(ep.get_variables_data()["x"]["r"])
        res = orc.ep_chain([orc.GBPriorFactor(N, rho=rho), lin, orc.GaussLikFactor(Y[0], nv)], [N, M], max_iter=n_it, damping=damping, tol=0.0)
        errs.append(float(np.abs(r[0] - res["r"][0]).max() / np.abs(res["r"][0]).max()))
    print("batch", batch, "damping", damping, "rel err after 1,2,3,12 it:", ["%.1e" % e for e in errs], ctx.prior_hint_stats(), flush=True)
'''
for env in ({}, {"TRAMP_PRIOR_FUSE": "0"}, {"TRAMP_EW_TILE": "0"}, {"TRAMP_PRIOR_FUSE": "0", "TRAMP_EW_TILE": "0"},
            {"TRAMP_ROTATE_OUTPUT": "0"}):
    print("== env", env, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env))
