"""Launch the componentwise updates added in round 6 twice each at 64 x 1e6 (f64 and f32), for the rocprofv3 --pmc passes of
tools/profile_r06.sh (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU ...): python tools/pmc_ew_r6.py [B] [N]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = T.default_context()
for np_dt in (np.float64, np.float32):
    bx = ctx.randn((B, N), np_dt, seed=2, scale=1.5)
    ax = ctx.scalars(0.7, B)
    a_new = ctx.empty((B,))
    b_new = ctx.empty((B, N), np_dt)
    kw = dict(out_a=a_new, out_b=b_new)
    n = N // 2
    yz = ctx.array(np.abs(np.random.RandomState(0).standard_normal((B, n)) + 1j * np.random.RandomState(1).standard_normal((B, n))).astype(np_dt), np_dt)
    facs = [T.PositivePrior(size=N, dtype=np_dt, ctx=ctx), T.ExponentialPrior(size=N, mean=0.7, dtype=np_dt, ctx=ctx),
            T.GaussianMixturePrior(size=N, dtype=np_dt, ctx=ctx),
            T.GaussianMixturePrior(size=N, probs=[0.2, 0.5, 0.3], means=[-2.0, 0.0, 1.5], vars=[0.3, 0.01, 2.0], dtype=np_dt, ctx=ctx)]
    ml = T.ModulusLikelihood(y=yz, dtype=np_dt, ctx=ctx)
    for _ in range(2):
        for f in facs:
            f.compute_forward_message(ax, bx, **kw)
        ml.compute_backward_message(ax, bx, **kw)
    del bx, b_new, yz
ctx.sync()
print("done")
