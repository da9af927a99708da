import sys, numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
from oracle import tramp_oracle as orc
ctx = T.default_context()
N, M, rho, nv = 4352, 192, 0.05, 1e-2
rs = np.random.RandomState(3)
W = orc.gaussian_ensemble(rs, M, N)
X = np.stack([orc.gb_sample(rs, N, rho) for _ in range(3)])
Y3 = X @ W.T + np.sqrt(nv) * rs.standard_normal((3, M))
lin = orc.LinearFactor(W)
rec = {}
def record(it, fwd, bwd, rs_, vs):
    rec[it] = dict(afx=fwd[0][0], abx=bwd[0][0], afz=fwd[1][0], abz=bwd[1][0], r=rs_[0].copy(), v=vs[0])
orc.ep_chain([orc.GBPriorFactor(N, rho=rho), lin, orc.GaussLikFactor(Y3[1], nv)], [N, M], max_iter=12, damping=None, tol=0.0, record=record)
model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ T.LinearChannel(W, ctx=ctx, svd="host")
         @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=Y3[1], var=nv, ctx=ctx)).to_model()
ep = T.ExpectationPropagation(model, ctx=ctx)
for n_it in range(1, 13):
    ep.iterate(max_iter=n_it, callback=T.PassCallback(), damping=None)
    d = ep.get_variables_data()["x"]; e = ep.get_edges_data(); o = rec[n_it - 1]
    print(n_it, "err r %.1e" % (np.abs(d["r"] - o["r"]).max() / np.abs(o["r"]).max()), "v", d["v"], o["v"],
          "a fwd x", float(e[("x", "fwd")][0]), o["afx"], "a bwd x", float(e[("x", "bwd")][0]), o["abx"], flush=True)
