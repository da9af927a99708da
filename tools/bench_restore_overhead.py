"""What the per-iteration message snapshot of restore_messages (the reference's "restoring old message dag") costs on a
matrix-free model: sparse deconvolution EP at N = 10^6 x 64 with the device stopping rule, snapshot on / off."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T  # noqa: E402

ctx = T.default_context()
n, B, rho, nv = 1_000_000, 64, 0.05, 1e-3
rs = np.random.RandomState(3)
ch = T.ConvChannel(rs.randn(n) / np.sqrt(n), ctx=ctx)
X = (rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n)))
Y = ch.convolve(ctx.array(X)).get() + np.sqrt(nv) * rs.standard_normal((B, n))
for restore in (False, True, False, True):
    model = (T.GaussBernoulliPrior(size=n, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ ch @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=3, callback=T.EarlyStoppingEP(tol=1e-30), damping=0.1, check_every=10, restore_messages=restore)
    ctx.sync()
    K = 20
    t0 = time.perf_counter()
    ep.iterate(max_iter=K, callback=T.EarlyStoppingEP(tol=1e-30), damping=0.1, warm_start=True, check_every=10,
               restore_messages=restore)
    ctx.sync()
    print(f"restore_messages={restore}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per iteration")
