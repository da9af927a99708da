#!/bin/bash
# rocm-smi power / clock samples while the two matrix passes of the headline shape loop (50000 x 100000, f64).
# Evidence for DESIGN.md section 4.1 (where the batch-64 passes sit against the package power limit).
# Usage: tools/power_log.sh > profiles/r02/power_log_N100000.txt
N=${1:-100000}
for B in 64 16; do
python3 - $N $B <<'PY' &
import sys, time; sys.path.insert(0,'.')
import numpy as np, tramp_amd as T
N=int(sys.argv[1]); B=int(sys.argv[2]); M=N//2
ctx=T.default_context(); dt=T.device.dtype_code(np.float64)
Vt=ctx.randn((M,N),np.float64,seed=1,scale=N**-0.5); x=ctx.randn((B,N),np.float64,seed=2); t=ctx.empty((B,M),np.float64)
y=ctx.randn((B,M),np.float64,seed=3); r=ctx.empty((B,N),np.float64)
for name,fn in (("NT", lambda: ctx._call("tramp_matvec_nt", dt, 0, M, N, B, Vt.ptr, x.ptr, t.ptr)),
                ("NN", lambda: ctx._call("tramp_matvec_nn", dt, 0, M, N, B, Vt.ptr, y.ptr, r.ptr))):
    fn(); ctx.sync()
    t0=time.time(); n=0
    while time.time()-t0<7:
        for _ in range(10): fn()
        ctx.sync(); n+=10
    el=time.time()-t0
    print(f"[workload] pass {name} R={M} C={N} batch {B} f64: {el/n*1e3:.3f} ms per launch, {2.0*M*N*B*n/el/1e12:.1f} TFLOP/s, matrix stream {M*N*8*n/el/1e12:.2f} TB/s", flush=True)
PY
PID=$!
sleep 6
for i in $(seq 1 16); do
  echo "[sample batch=$B t=$i] $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level|fclk clock level' | sed -E 's/^GPU\[0\]\s*:\s*//' | tr '\n' ';')"
  sleep 0.5
done
wait $PID
done
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|cap" | head -4
