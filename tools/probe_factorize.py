"""GPU probe: time tramp_linear_factorize (rocBLAS syrk + rocSOLVER syevd + gemm) vs N."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
print(ctx.device_name(), [x / 2**30 for x in ctx.mem_info()])
for N in [int(a) for a in sys.argv[1:]]:
    M = N // 2
    W = ctx.randn((M, N), np.float64, seed=1, scale=1 / np.sqrt(N))
    ctx.sync()
    t0 = time.perf_counter()
    lin = T.LinearChannel(W, ctx=ctx, svd="device")
    ctx.sync()
    dt = time.perf_counter() - t0
    msg = f"N={N} M={M} factorize {dt:.2f}s rank={lin.rank} smax={lin.spectrum.max()**.5:.4f} smin={lin.spectrum[:M].min()**.5:.4f}"
    if N <= 4096:
        Wh = W.get()
        U, s, Vt = lin._Uk.get(), lin._sing.get(), lin._Vt.get()
        rec = np.abs((U * s) @ Vt - Wh).max()
        orth = np.abs(Vt @ Vt.T - np.eye(M)).max()
        sh = np.linalg.svd(Wh, compute_uv=False)
        msg += f" recon={rec:.2e} orthV={orth:.2e} sing_err={np.abs(s - sh).max():.2e}"
    print(msg, flush=True)
    del lin, W
