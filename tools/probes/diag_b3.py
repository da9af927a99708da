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
refs = {}
for n_it in (1, 2, 3, 4, 6, 12):
    refs[n_it] = orc.ep_chain([orc.GBPriorFactor(N, rho=rho), lin, orc.GaussLikFactor(Y3[1], nv)], [N, M], max_iter=n_it, damping=None, tol=0.0)
# the hard instance (index 1) placed at different positions of ensembles of different sizes
for batch, pos in ((1, 0), (2, 1), (3, 1), (3, 0), (3, 2), (4, 1), (5, 3), (8, 1)):
    Y = np.repeat(Y3[:1], batch, axis=0); Y[pos] = Y3[1]
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ T.LinearChannel(W, ctx=ctx, svd="host")
             @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=Y if batch > 1 else Y[0], var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    out = []
    for n_it in (1, 2, 3, 4, 6, 12):
        ep.iterate(max_iter=n_it, callback=T.PassCallback(), damping=None)
        r = np.atleast_2d(ep.get_variables_data()["x"]["r"])[pos]
        out.append("%.1e" % (np.abs(r - refs[n_it]["r"][0]).max() / np.abs(refs[n_it]["r"][0]).max()))
    print("batch", batch, "position", pos, "rel err after 1,2,3,4,6,12 it", out, flush=True)
