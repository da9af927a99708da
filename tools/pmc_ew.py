"""Launch every componentwise factor update once or twice at 64 x 1e6 (f64 and f32) on synthetic messages, for
the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU ...): python tools/pmc_ew.py [B] [N]
Includes the producer-fused pair (convolution channel's inverse FFT leaves the prior's moments, the prior
update is one pass)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ONLY_K7 = len(sys.argv) > 3 and sys.argv[3] == "k7"
ctx = T.default_context()
for np_dt in (np.float64, np.float32):
    bz = ctx.randn((B, N), np_dt, seed=1, scale=1.5); bx = ctx.randn((B, N), np_dt, seed=2, scale=1.5)
    y = ctx.array(np.sign(np.random.RandomState(0).standard_normal((B, N))).astype(np_dt), np_dt)
    az = ctx.scalars(1.3, B); ax = ctx.scalars(0.7, B)
    a_new = ctx.empty((B,)); b_new = ctx.empty((B, N), np_dt)
    kw = dict(out_a=a_new, out_b=b_new)
    gb = T.GaussBernoulliPrior(size=N, rho=0.05, dtype=np_dt, ctx=ctx)
    l1 = T.MAP_L1NormPrior(size=N, gamma=1.0, dtype=np_dt, ctx=ctx)
    sg = T.SgnLikelihood(y=y, dtype=np_dt, ctx=ctx)
    ab = T.AbsLikelihood(y=y, dtype=np_dt, ctx=ctx)
    relu, sgn, abs_ = T.ReluChannel(dtype=np_dt, ctx=ctx), T.SgnChannel(dtype=np_dt, ctx=ctx), T.AbsChannel(dtype=np_dt, ctx=ctx)
    bp = T.BinaryPrior(size=N, p_pos=0.3, dtype=np_dt, ctx=ctx)
    yr = ctx.array(np.maximum(np.random.RandomState(1).standard_normal((B, N)), 0.0).astype(np_dt), np_dt)
    rl = T.ReluLikelihood(y=yr, dtype=np_dt, ctx=ctx)
    for _ in range(2):
        if not ONLY_K7:
            gb.compute_forward_message(ax, bx, **kw)
            l1.compute_forward_message(ax, bx, **kw)
            bp.compute_forward_message(ax, bx, **kw)
            rl.compute_backward_message(az, bz, **kw)
            sg.compute_backward_message(az, bz, **kw)
            ab.compute_backward_message(az, bz, **kw)
        relu.compute_backward_message(az, bz, ax, bx, **kw)
        relu.compute_forward_message(az, bz, ax, bx, **kw)
        sgn.compute_backward_message(az, bz, ax, bx, **kw)
        abs_.compute_backward_message(az, bz, ax, bx, **kw)
    if N % 2 == 0 and not ONLY_K7:
        # fused pair: producer (inverse FFT with the message update) + single-pass prior
        ch = T.ConvChannel(np.random.RandomState(3).randn(N) / np.sqrt(N), dtype=np_dt, ctx=ctx)
        a_msg, b_msg = ctx.empty((B,)), ctx.empty((B, N), np_dt)
        k = gb.moment_hint()
        for _ in range(2):
            ctx._call("tramp_prior_hint", k[0], k[1], k[2], k[3], N, B, b_msg.ptr, a_msg.ptr)
            ch.compute_backward_message(az, bz, ax, bx, out_a=a_msg, out_b=b_msg)
            gb.compute_forward_message(a_msg, b_msg, **kw)
        del ch
    del bz, bx, y, yr, b_new
ctx.sync()
print("done", ctx.prior_hint_stats())
