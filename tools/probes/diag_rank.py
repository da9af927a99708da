import sys, numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
np.set_printoptions(linewidth=200, precision=3)
for shape in ((60, 90), (90, 60)):
    M, N = shape; k = min(M, N)
    rs = np.random.RandomState(11)
    U, _ = np.linalg.qr(rs.randn(M, k)); V, _ = np.linalg.qr(rs.randn(N, k))
    s = np.logspace(0, -12, k); s[-3:] = 0.0
    W = (U * s) @ V.T
    ch = T.LinearChannel(W, ctx=ctx, svd="device")
    sd = ch._sing.get(); sr = np.linalg.svd(W, compute_uv=False)
    print(shape, "rank", ch.rank, np.linalg.matrix_rank(W), "tol numpy", sr[0] * max(M, N) * 2.2e-16)
    print(" device tail", sd[-8:]); print(" lapack tail", sr[-8:])
    print(" max |s^2 - ref^2|", np.abs(sd**2 - sr**2).max())
# per-instance EP check, batch 3 undamped
from oracle import tramp_oracle as orc
rs = np.random.RandomState(3)
N, M, rho, nv = 4352, 192, 0.05, 1e-2
W = orc.gaussian_ensemble(rs, M, N)
for batch, damping in ((3, None), (3, 0.1), (64, None)):
    rs2 = np.random.RandomState(3); orc.gaussian_ensemble(rs2, M, N)
    X = np.stack([orc.gb_sample(rs2, N, rho) for _ in range(batch)])
    Y = X @ W.T + np.sqrt(nv) * rs2.standard_normal((batch, M))
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ T.LinearChannel(W, ctx=ctx, svd="host")
             @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=12, callback=T.PassCallback(), damping=damping)
    r = np.atleast_2d(ep.get_variables_data()["x"]["r"]); v = np.atleast_1d(ep.get_variables_data()["x"]["v"])
    lin = orc.LinearFactor(W)
    for b in range(3):
        res = orc.ep_chain([orc.GBPriorFactor(N, rho=rho), lin, orc.GaussLikFactor(Y[b], nv)], [N, M], max_iter=12, damping=damping, tol=0.0)
        print("batch", batch, "damping", damping, "instance", b, "rel err r %.2e" % (np.abs(r[b] - res["r"][0]).max() / np.abs(res["r"][0]).max()), "v", v[b], res["v"][0])
