"""Worker of tests/test_gpu_coop.py::test_one_launch_form_gives_up_cleanly (its environment withholds one block sum from the
one-launch update and shortens the kernel's deadline): prints one JSON line of what happened."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T                                                    # noqa: E402
from oracle import tramp_oracle as orc                                   # noqa: E402

ctx = T.default_context(0)
B, n = 3, 200_000
rs = np.random.RandomState(2)
ax = rs.uniform(0.5, 2.0, B)
bx = rs.standard_normal((B, n))
p = T.GaussBernoulliPrior(size=n, rho=0.1, ctx=ctx)
out = {}
runs0 = ctx.coop_runs()
a1, b1 = p.compute_forward_message(ctx.scalars(ax), ctx.array(bx))
a1, b1 = np.array(a1), np.array(b1)
out["coop_ran"] = ctx.coop_runs() == runs0 + 1
out["gave_up"] = ctx.resident_gave_up()
runs1 = ctx.coop_runs()
a2, b2 = p.compute_forward_message(ctx.scalars(ax), ctx.array(bx))       # the context is on the tile kernels now
a2, b2 = np.array(a2), np.array(b2)
out["second_call_on_tiles"] = ctx.coop_runs() == runs1
out["nan_instances_first"] = np.isnan(a1).astype(int).tolist()
out["nan_instances_second"] = int(np.isnan(a2).sum())
ok = ~np.isnan(a1)
out["finite_agree"] = bool(np.allclose(a1[ok], a2[ok], rtol=1e-11) and np.allclose(b1[ok], b2[ok], rtol=1e-9, atol=1e-9))
ref = [orc.gb_prior_message(ax[i], bx[i], rho=0.1) for i in range(B)]
out["oracle"] = bool(all(abs(a2[i] - ref[i][0]) <= 1e-9 * abs(ref[i][0]) and np.allclose(b2[i], ref[i][1], rtol=1e-9, atol=1e-9)
                         for i in range(B)))
print(json.dumps(out))
