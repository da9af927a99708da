import numpy as np, sys
sys.path.insert(0, '.')
import tramp_amd as T
from tests.conftest import load_golden
g = load_golden("factors")
ctx = T.default_context()
out = {}
cls = dict(sgn=T.SgnChannel, abs=T.AbsChannel, relu=T.ReluChannel)
for i, c in enumerate(g["pwl"]):
    if c["az"] == 1e-11:
        ch = cls[c["kind"]](ctx=ctx)
        for d, post, msg in (("fwd", ch.compute_forward_posterior, ch.compute_forward_message),
                             ("bwd", ch.compute_backward_posterior, ch.compute_backward_message)):
            r, v = post(c["az"], c["bz"], c["ax"], c["bx"])
            a, b = msg(c["az"], c["bz"], c["ax"], c["bx"])
            out[f"{i}_{d}_r"] = r; out[f"{i}_{d}_v"] = v; out[f"{i}_{d}_a"] = a; out[f"{i}_{d}_b"] = b
np.savez("gpurun_out/dbg_pwl.npz", **out)
