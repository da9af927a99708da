import sys, numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
B, N = 64, 1000000
bx = ctx.randn((B, N), np.float64, seed=2, scale=1.5)
ax = ctx.scalars(0.7, B)
a_new = ctx.empty((B,)); b_new = ctx.empty((B, N), np.float64)
for name, pr in (("exp", T.ExponentialPrior(size=N, mean=0.7, ctx=ctx)), ("pos", T.PositivePrior(size=N, ctx=ctx))):
    for i in range(40):
        r0 = ctx.resident_runs()
        e0, e1 = ctx.event(), ctx.event()
        e0.record()
        pr.compute_forward_message(ax, bx, out_a=a_new, out_b=b_new)
        e1.record()
        ms = e0.elapsed_ms(e1)
        a = a_new.get()
        if i % 3 == 0: ctx.sync()
        print(name, i, "resident", ctx.resident_runs() - r0, "ms", round(ms, 3), "nan a_new", int(np.isnan(a).sum()), flush=True)
