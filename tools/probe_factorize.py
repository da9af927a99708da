"""GPU probe: time tramp_linear_factorize (Gram + tramp_eigh + gemm) vs N and check the factors.
    python tools/probe_factorize.py 100000            # BASELINE configs[3]: k = 50000, spectral split
    TRAMP_EIG_CAP=16384 python tools/probe_factorize.py 65536
Checks that need no host copy of the factors: Vt Vt^T = I on random probe vectors, the Gram identity
W^T (W x) = Vt^T S^2 Vt x, the Marchenko-Pastur edges of the spectrum."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
print(ctx.device_name(), [x / 2**30 for x in ctx.mem_info()], flush=True)
for N in [int(a) for a in sys.argv[1:]]:
    M = N // 2
    W = ctx.randn((M, N), np.float64, seed=1, scale=1 / np.sqrt(N))
    ctx.sync()
    t0 = time.perf_counter()
    lin = T.LinearChannel(W, ctx=ctx, svd="device", keep_W=True)
    ctx.sync()
    dt = time.perf_counter() - t0
    s = lin._sing.get()
    msg = (f"N={N} M={M} factorize {dt:.2f}s rank={lin.rank} smax={s.max():.6f} (MP {1 + np.sqrt(.5):.6f}) "
           f"smin={s.min():.6f} (MP {1 - np.sqrt(.5):.6f}) sum s^2/M={np.sum(s**2) / M:.8f}")
    # probe vectors through the C ABI mat-vecs
    B = 4
    dt_code = T.device.dtype_code(np.float64)
    x = ctx.randn((B, N), np.float64, seed=7)
    t = ctx.empty((B, M), np.float64)
    ctx._call("tramp_matvec_nt", dt_code, 0, M, N, B, lin._Vt.ptr, x.ptr, t.ptr)          # t = Vt x
    back = ctx.empty((B, N), np.float64)
    ctx._call("tramp_matvec_nn", dt_code, 0, M, N, B, lin._Vt.ptr, t.ptr, back.ptr)       # Vt^T t
    t2 = ctx.empty((B, M), np.float64)
    ctx._call("tramp_matvec_nt", dt_code, 0, M, N, B, lin._Vt.ptr, back.ptr, t2.ptr)      # Vt Vt^T t = t
    th, t2h = t.get(), t2.get()
    msg += f" |VtVt^T t - t|/|t|={np.abs(t2h - th).max() / np.abs(th).max():.2e}"
    # W^T W x  vs  Vt^T S^2 Vt x
    z = ctx.empty((B, M), np.float64)
    ctx._call("tramp_matvec_nt", dt_code, 0, M, N, B, W.ptr, x.ptr, z.ptr)
    g1 = ctx.empty((B, N), np.float64)
    ctx._call("tramp_matvec_nn", dt_code, 0, M, N, B, W.ptr, z.ptr, g1.ptr)
    ts = ctx.array(th * s[None, :]**2, np.float64)
    g2 = ctx.empty((B, N), np.float64)
    ctx._call("tramp_matvec_nn", dt_code, 0, M, N, B, lin._Vt.ptr, ts.ptr, g2.ptr)
    g1h, g2h = g1.get(), g2.get()
    msg += f" |W^TWx - VS^2V^Tx|/|.|={np.abs(g1h - g2h).max() / np.abs(g1h).max():.2e}"
    if N <= 4096:
        Wh = W.get()
        U, Vt = lin._Uk.get(), lin._Vt.get()
        rec = np.abs((U * s) @ Vt - Wh).max()
        orth = np.abs(Vt @ Vt.T - np.eye(M)).max()
        sh = np.linalg.svd(Wh, compute_uv=False)
        msg += f" recon={rec:.2e} orthV={orth:.2e} sing_err={np.abs(s - sh).max():.2e}"
    print(msg, flush=True)
    del lin, W
