"""Two-pass 2^a 5^b FFT (csrc/fft_pow2.hip) in isolation: python tools/bench_fft.py [h | log2 h] [batch] [f64|f32]"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
q = int(sys.argv[1]) if len(sys.argv) > 1 else 19
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
np_dt = np.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else np.float64
ctx = T.default_context()
h = (1 << q) if q < 64 else q
x = ctx.randn((B, 2 * h), np_dt, seed=1)
y = ctx.empty((B, 2 * h), np_dt)
f, i = C.c_double(), C.c_double()
ctx._call("tramp_fft_bench", T.device.dtype_code(np_dt), h, B, 10, x.ptr, y.ptr, C.byref(f), C.byref(i))
byt = 2.0 * B * h * 2 * np.dtype(np_dt).itemsize            # one read + one write of the array
print(f"h={h} batch={B} {np_dt.__name__}: forward {f.value:.3f} ms ({byt/f.value/1e6:.0f} GB/s of read+write once), "
      f"inverse {i.value:.3f} ms ({byt/i.value/1e6:.0f} GB/s)")
