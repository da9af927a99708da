"""Time the componentwise updates added in round 6 in isolation (as tools/bench_ew.py): python tools/bench_ew_r6.py [B] [N]"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = T.default_context()


def timeit(fn, reps):
    for _ in range(2):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:          # warm launches: the clocks ramp over tens of milliseconds
        for _ in range(5):
            fn()
        ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    return e0.elapsed_ms(e1) / reps


for np_dt in (np.float64, np.float32):
    es = np.dtype(np_dt).itemsize
    bx = ctx.randn((B, N), np_dt, seed=2, scale=1.5)
    ax = ctx.scalars(0.7, B)
    a_new = ctx.empty((B,))
    b_new = ctx.empty((B, N), np_dt)
    kw = dict(out_a=a_new, out_b=b_new)
    ex = T.ExponentialPrior(size=N, mean=0.7, dtype=np_dt, ctx=ctx)
    po = T.PositivePrior(size=N, dtype=np_dt, ctx=ctx)
    g2 = T.GaussianMixturePrior(size=N, dtype=np_dt, ctx=ctx)
    g3 = T.GaussianMixturePrior(size=N, probs=[0.2, 0.5, 0.3], means=[-2.0, 0.0, 1.5], vars=[0.3, 0.01, 2.0], dtype=np_dt, ctx=ctx)
    g8 = T.GaussianMixturePrior(size=N, probs=[0.125] * 8, means=list(np.linspace(-3, 3, 8)), vars=[0.2] * 8, dtype=np_dt, ctx=ctx)
    cases = [("exp_prior", 2, lambda: ex.compute_forward_message(ax, bx, **kw)),
             ("positive_prior", 2, lambda: po.compute_forward_message(ax, bx, **kw)),
             ("mixture_K2", 2, lambda: g2.compute_forward_message(ax, bx, **kw)),
             ("mixture_K3", 2, lambda: g3.compute_forward_message(ax, bx, **kw)),
             ("mixture_K8", 2, lambda: g8.compute_forward_message(ax, bx, **kw))]
    # modulus likelihood: N/2 complex components per instance (the same bytes of message as the priors above)
    n = N // 2
    yz = ctx.array(np.abs(np.random.RandomState(0).standard_normal((B, n)) + 1j * np.random.RandomState(1).standard_normal((B, n))).astype(np_dt), np_dt)
    ml = T.ModulusLikelihood(y=yz, dtype=np_dt, ctx=ctx)
    cases.append(("modulus_lik", 2.5, lambda: ml.compute_backward_message(ax, bx, **kw)))
    for name, nT, fn in cases:
        ms = timeit(fn, 5)
        alg = nT * es * N * B
        print(f"{name:15s} {np_dt.__name__} B={B} N={N}: {ms*1e3:9.1f} us  {alg/ms/1e6:7.0f} GB/s algorithmic ({alg/ms/1e6/8000*100:4.1f}% of 8 TB/s)", flush=True)
    del bx, b_new, yz
