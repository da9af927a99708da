import sys, gc, numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
def free_mb():
    ctx.sync(); gc.collect()
    f, t = ctx.mem_info()
    return f / 2**20
rs = np.random.RandomState(0)
N, M, B = 4608, 2304, 8
W = rs.randn(M, N) / np.sqrt(N)
def one(i):
    X = rs.standard_normal((B, N)) * rs.binomial(1, 0.1, (B, N))
    Y = X @ W.T + 0.1 * rs.standard_normal((B, M))
    ch = T.LinearChannel(W, ctx=ctx, svd="device" if i % 2 else "host")
    model = (T.GaussBernoulliPrior(size=N, rho=0.1, ctx=ctx) @ T.SISOVariable(id="x") @ ch @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=Y, var=1e-2, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=30, damping=0.1, graph=bool(i % 3 == 0))
    cv = T.ConvChannel(rs.randn(16384) / 128, ctx=ctx)
    m2 = (T.GaussBernoulliPrior(size=16384, rho=0.1, ctx=ctx) @ T.SISOVariable(id="x") @ cv @ T.SISOVariable(id="z")
          @ T.SgnLikelihood(y=np.sign(rs.randn(2, 16384)), ctx=ctx)).to_model()
    T.ExpectationPropagation(m2, ctx=ctx).iterate(max_iter=5, damping=0.2)
    return float(np.mean(ep.get_variables_data()["x"]["v"]))
one(0); one(1)
f0 = free_mb()
for i in range(2, 42):
    one(i)
    if i % 10 == 1: print(i, "free MiB", round(free_mb(), 1), "delta", round(free_mb() - f0, 1), flush=True)
print("final delta MiB", round(free_mb() - f0, 1))
