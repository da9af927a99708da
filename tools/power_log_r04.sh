#!/bin/bash
# rocm-smi power / clock samples while (a) the f64 pass pair and (b) the 7-digit pass pair of the headline shape loop
# (50000 x 100000 x 64).  Usage: tools/power_log_r04.sh > profiles/r04/power_log_f64_and_digits.txt
for engine in f64 digits7; do
python3 - $engine <<'PY' &
import sys, time; sys.path.insert(0,'.')
import numpy as np, tramp_amd as T
engine = sys.argv[1]
N, B = 100000, 64; M = N // 2
ctx = T.default_context(); dt = T.device.dtype_code(np.float64)
Vt = ctx.randn((M, N), np.float64, seed=1, scale=N ** -0.5); x = ctx.randn((B, N), np.float64, seed=2); y = ctx.randn((B, M), np.float64, seed=3)
t = ctx.empty((B, M), np.float64); r = ctx.empty((B, N), np.float64)
algo = 2
if engine != "f64":
    ctx.attach_digits(Vt, 7, 3); algo = 3
def pair():
    ctx._call("tramp_matvec_nt", dt, algo, M, N, B, Vt.ptr, x.ptr, t.ptr)
    ctx._call("tramp_matvec_nn", dt, algo, M, N, B, Vt.ptr, y.ptr, r.ptr)
pair(); ctx.sync()
print(f"[workload {engine}] started", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < 12:
    for _ in range(10): pair()
    ctx.sync(); n += 10
el = time.time() - t0
print(f"[workload {engine}] pass pair: {el / n * 1e3:.3f} ms, {4.0 * M * N * B * n / el / 1e12:.1f} TFLOP/s (f64-equivalent)", flush=True)
PY
PID=$!
sleep 9
for i in $(seq 1 10); do
  echo "[sample $engine t=$i] $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level' | sed -E 's/^GPU\[0\]\s*:\s*//' | tr '\n' ';')"
  sleep 0.6
done
wait $PID
done
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|cap" | head -4
