"""The bench line's structured_2d workload alone (for rocprofv3 --kernel-trace --stats): sparse deconvolution EP on
1024 x 1024 images x 64, 20 iterations."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T  # noqa: E402

ctx = T.default_context()
shape, B, rho, nv = (1024, 1024), 64, 0.05, 1e-3
n = shape[0] * shape[1]
rs = np.random.RandomState(4)
X = (rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n)))
ch = T.ConvChannel(rs.standard_normal(shape) / np.sqrt(n), ctx=ctx)
Y = ch.convolve(ctx.array(X)).get() + np.sqrt(nv) * rs.standard_normal((B, n))
model = (T.GaussBernoulliPrior(size=shape, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ ch @ T.SISOVariable(id="z")
         @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
ep = T.ExpectationPropagation(model, ctx=ctx)
ep.iterate(max_iter=23, callback=T.PassCallback(), damping=0.1, check_every=10)
ctx.sync()
print("mse", float(np.mean(ep.mse("x", X))))
