"""VERDICT r1 item 3, measured: the two matrix passes of the headline shape done per ROW CHUNK of Vt (a chunk that
fits the 256 MiB Infinity Cache: NT on the chunk, then NN on the same chunk, so that the second read could hit the
cache) against the whole-matrix passes.  Uses the library's own kernels on a chunk-shaped operand; the chunked
variant is charged NOTHING for the accumulation of r across chunks (each chunk's NN writes its own output), i.e. this
is an upper bound for a chunked schedule built from these kernels.
    python tools/probes/probe_chunked_passes.py [N] [B] [rows_per_chunk ...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
chunks = [int(a) for a in sys.argv[3:]] or [160, 256, 320, 1024, 4096]
M = N // 2
ctx = T.default_context()
dt = T.device.dtype_code(np.float64)
Vt = ctx.randn((M, N), np.float64, seed=1, scale=N ** -0.5)
x = ctx.randn((B, N), np.float64, seed=2)
y = ctx.randn((B, M), np.float64, seed=3)
t = ctx.empty((B, M), np.float64)
r = ctx.empty((B, N), np.float64)


def timed(fn, reps=3):
    fn(); ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    return e0.elapsed_ms(e1) / reps


def whole():
    ctx._call("tramp_matvec_nt", dt, 0, M, N, B, Vt.ptr, x.ptr, t.ptr)
    ctx._call("tramp_matvec_nn", dt, 0, M, N, B, Vt.ptr, y.ptr, r.ptr)


ms = timed(whole)
print(f"whole matrix  {M} x {N}, batch {B}: {ms:8.3f} ms per NT+NN pair, {4.0*M*N*B/ms/1e9:6.1f} TFLOP/s", flush=True)
es = 8
for R in chunks:
    nchunk = (M + R - 1) // R
    tc = ctx.empty((B, R), np.float64)
    yc = ctx.randn((B, R), np.float64, seed=5)

    def chunked():
        for c in range(nchunk):
            r0 = min(c * R, M - R)
            p = Vt.ptr + r0 * N * es
            ctx._call("tramp_matvec_nt", dt, 0, R, N, B, p, x.ptr, tc.ptr)
            ctx._call("tramp_matvec_nn", dt, 0, R, N, B, p, yc.ptr, r.ptr)

    ms_c = timed(chunked, reps=2)
    print(f"chunks of {R:5d} rows ({R*N*es/2**20:6.0f} MiB, {nchunk:4d} chunk pairs): {ms_c:8.3f} ms per NT+NN sweep, "
          f"{4.0*M*N*B/ms_c/1e9:6.1f} TFLOP/s", flush=True)
