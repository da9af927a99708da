"""Half-line priors at 64 x 1e6 f64: per-call time of the lean resident form + redo of the flagged instances, with and
without a host sync between calls (the redo's pinned-host statistics are read without synchronising)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
B, N = 64, 1000000
bx = ctx.randn((B, N), np.float64, seed=2, scale=1.5)
ax = ctx.scalars(0.7, B)
a_new = ctx.empty((B,)); b_new = ctx.empty((B, N), np.float64)
for name, pr in (("exp", T.ExponentialPrior(size=N, mean=0.7, ctx=ctx)), ("pos", T.PositivePrior(size=N, ctx=ctx))):
    for mode in ("sync each", "back to back"):
        for _ in range(30):
            pr.compute_forward_message(ax, bx, out_a=a_new, out_b=b_new)
            if mode == "sync each":
                ctx.sync()
        ctx.sync()
        r0 = ctx.resident_runs()
        e0, e1 = ctx.event(), ctx.event()
        e0.record()
        for _ in range(20):
            pr.compute_forward_message(ax, bx, out_a=a_new, out_b=b_new)
            if mode == "sync each":
                ctx.sync()
        e1.record()
        print(name, mode, "resident runs", ctx.resident_runs() - r0, "of 20; us per call", round(e0.elapsed_ms(e1) * 50, 1), flush=True)
