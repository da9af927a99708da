"""Time the componentwise factor updates in isolation: python tools/bench_ew.py [B] [N]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = T.default_context()
WARM_S = float(__import__("os").environ.get("BENCH_EW_WARM_S", "0.05"))
def timeit(fn, reps):
    for _ in range(2): fn()
    ctx.sync()
    # ~50 ms of the same launches first: the device idles while Python builds the next factor, and its clocks take tens of
    # milliseconds to come back up - seven calls of a 0.4 ms instruction-bound update are over before they have
    # (BENCH_EW_WARM_S=0: the cold-start figure)
    t0 = __import__("time").perf_counter()
    while __import__("time").perf_counter() - t0 < WARM_S:
        for _ in range(10): fn()
        ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(reps): fn()
    e1.record()
    return e0.elapsed_ms(e1) / reps
for np_dt in (np.float64, np.float32):
    es = np.dtype(np_dt).itemsize
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
    cases = [("gb_prior", 2, lambda: gb.compute_forward_message(ax, bx, **kw)),
             ("l1_prior", 2, lambda: l1.compute_forward_message(ax, bx, **kw)),
             ("binary_prior", 2, lambda: bp.compute_forward_message(ax, bx, **kw)),
             ("relu_lik", 3, lambda: rl.compute_backward_message(az, bz, **kw)),
             ("sgn_lik", 3, lambda: sg.compute_backward_message(az, bz, **kw)),
             ("abs_lik", 3, lambda: ab.compute_backward_message(az, bz, **kw)),
             ("relu_bwd", 3, lambda: relu.compute_backward_message(az, bz, ax, bx, **kw)),
             ("relu_fwd", 3, lambda: relu.compute_forward_message(az, bz, ax, bx, **kw)),
             ("sgn_bwd", 3, lambda: sgn.compute_backward_message(az, bz, ax, bx, **kw)),
             ("abs_bwd", 3, lambda: abs_.compute_backward_message(az, bz, ax, bx, **kw))]
    for name, nT, fn in cases:
        ms = timeit(fn, 20 if B * N < 4e6 else 5)
        alg = nT * es * N * B
        print(f"{name:12s} {np_dt.__name__} B={B} N={N}: {ms*1e3:9.1f} us  {alg/ms/1e6:7.0f} GB/s algorithmic ({alg/ms/1e6/8000*100:4.1f}% of 8 TB/s)  {N*B/ms/1e6:8.1f} Gcomp/s", flush=True)
    del bz, bx, y, yr, b_new
