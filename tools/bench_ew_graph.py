"""One instance of 1e6 components (BASELINE configs[1] read literally): the Gauss-Bernoulli update replayed inside a hipGraph
(how the EP loop issues it with graph=True) against eager launches: python tools/bench_ew_graph.py [B] [N]"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = T.default_context()
for np_dt in (np.float64, np.float32):
    bx = ctx.randn((B, N), np_dt, seed=2, scale=1.5)
    ax = ctx.scalars(0.7, B)
    a_new = ctx.empty((B,)); b_new = ctx.empty((B, N), np_dt)
    gb = T.GaussBernoulliPrior(size=N, rho=0.05, dtype=np_dt, ctx=ctx)
    fn = lambda: gb.compute_forward_message(ax, bx, out_a=a_new, out_b=b_new)      # noqa: E731
    for _ in range(200): fn()
    ctx.sync()
    K = 50
    ctx._call("tramp_graph_begin")
    for _ in range(K): fn()
    g = C.c_void_p()
    ctx._call("tramp_graph_end", C.byref(g))
    for _ in range(20): ctx._call("tramp_graph_launch", g)
    ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(20): ctx._call("tramp_graph_launch", g)
    e1.record()
    t_graph = e0.elapsed_ms(e1) / (20 * K) * 1e3
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(20 * K): fn()
    e1.record()
    t_eager = e0.elapsed_ms(e1) / (20 * K) * 1e3
    print(f"gb_prior {np_dt.__name__} B={B} N={N}: {t_graph:.2f} us per update inside a replayed graph of {K}, {t_eager:.2f} us eager "
          f"(one-launch form ran: {ctx.coop_runs() > 0})", flush=True)
    ctx._call("tramp_graph_destroy", g)
