"""The message snapshot of restore_messages on a tree model (sparse-gradient EP, 4096 x 4096 x 4, f64): on / off."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T  # noqa: E402

side, B, nv, rho = 4096, 4, 1e-2, 0.02
shape, n = (side, side), side * side
ctx = T.default_context()
rs = np.random.RandomState(5)
X = np.kron(rs.standard_normal((B, 16, 16)), np.ones((side // 16, side // 16)))
Y = X + np.sqrt(nv) * rs.standard_normal(X.shape)
kw = dict(ctx=ctx)
model = (T.GaussianPrior(size=shape, **kw) @ T.SIMOVariable(id="x", n_next=2) @ (
    T.GaussianLikelihood(y=Y.reshape(B, n), var=nv, **kw) + (
        T.GradientChannel(shape=shape, **kw) + T.GaussBernoulliPrior(size=(2,) + shape, rho=rho, **kw))
    @ T.MILeafVariable(id="z", n_prev=2))).to_model()
init = T.ConstantInit(a=0.1, b=1.0)
for restore in (False, True, False, True):
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=3, callback=T.EarlyStoppingEP(tol=1e-30), damping=0.3, check_every=1 << 30, initializer=init,
               restore_messages=restore)
    ctx.sync()
    K = 15
    t0 = time.perf_counter()
    ep.iterate(max_iter=K, callback=T.EarlyStoppingEP(tol=1e-30), damping=0.3, warm_start=True, check_every=1 << 30,
               restore_messages=restore)
    ctx.sync()
    print(f"tree restore_messages={restore}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per iteration")
    del ep
