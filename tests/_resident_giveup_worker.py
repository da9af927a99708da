"""Worker of tests/test_gpu_resident.py::test_resident_gives_up_cleanly (its environment withholds one tile sum from the
resident pipeline and shortens the kernel's deadline): prints one JSON line of what happened."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramp_amd as T                                                    # noqa: E402
from oracle import tramp_oracle as orc                                   # noqa: E402

ctx = T.default_context(0)
cu = int(ctx.device_name().rsplit(",", 1)[1].split()[0])
tiles, comps = 7, 2048
n = tiles * comps
B = (4 * (2 * cu - 8) + 16 + tiles - 1) // tiles
rs = np.random.RandomState(1)
out = {"B": B, "n": n}
which = sys.argv[1]
if which == "relu":
    az, ax = rs.uniform(0.8, 2.0, B), rs.uniform(0.5, 1.5, B)
    bz, bx = rs.standard_normal((B, n)), rs.standard_normal((B, n))
    ch = T.ReluChannel(ctx=ctx)
    call = lambda: ch.compute_backward_message(ctx.scalars(az), ctx.array(bz), ctx.scalars(ax), ctx.array(bx))   # noqa: E731
else:
    ax = rs.uniform(0.5, 2.0, B)
    bx = rs.standard_normal((B, n))
    p = T.GaussBernoulliPrior(size=n, rho=0.1, ctx=ctx)
    call = lambda: p.compute_forward_message(ctx.scalars(ax), ctx.array(bx))                                       # noqa: E731
runs0 = ctx.resident_runs()
a1, b1 = call()
a1, b1 = np.array(a1), np.array(b1)
out["resident_ran"] = ctx.resident_runs() == runs0 + 1
out["gave_up"] = ctx.resident_gave_up()
runs1 = ctx.resident_runs()
a2, b2 = call()                                                          # the context is on the tile kernels now
a2, b2 = np.array(a2), np.array(b2)
out["second_call_on_tiles"] = ctx.resident_runs() == runs1
out["nan_instances_first"] = int(np.isnan(a1).sum())
out["nan_instances_second"] = int(np.isnan(a2).sum())
ok = ~np.isnan(a1)
out["finite_agree"] = bool(np.allclose(a1[ok], a2[ok], rtol=1e-11) and np.allclose(b1[ok], b2[ok], rtol=1e-9, atol=1e-9))
out["bitwise_equal_instances"] = int(sum(np.array_equal(b1[i], b2[i]) and a1[i] == a2[i] for i in range(B)))
if which == "relu":
    ref = orc.pwl_channel_message("relu", "bwd", az[0], bz[0], ax[0], bx[0])
else:
    ref = orc.gb_prior_message(ax[0], bx[0], rho=0.1)
out["oracle_instance0"] = bool(abs(a2[0] - ref[0]) <= 1e-9 * abs(ref[0]) and np.allclose(b2[0], ref[1], rtol=1e-9, atol=1e-9))
print(json.dumps(out))
