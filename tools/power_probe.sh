#!/bin/bash
# Samples rocm-smi power / clocks while a matvec benchmark loops (is the B=64 f64 plateau a power cap?)
MODE=${1:-16}
export TRAMP_MFMA_F64_SHAPE=$MODE
[ "$MODE" = "ring" ] && export TRAMP_MFMA_F64_SHAPE=4 TRAMP_MFMA_RING=1
[ "$MODE" = "ringdbg" ] && export TRAMP_MFMA_F64_SHAPE=4 TRAMP_MFMA_RING=1 TRAMP_MFMA_RING_DBG=1
python3 - <<'PY' &
import sys, time; sys.path.insert(0,'.')
import numpy as np, tramp_amd as T, os
npdt=np.float32 if os.environ.get("PB_DTYPE")=="f32" else np.float64
ctx=T.default_context(); dt=T.device.dtype_code(npdt)
N=65536; M=N//2; B=int(os.environ.get("PB_BATCH","64"))
Vt=ctx.randn((M,N),npdt,seed=1,scale=N**-0.5); x=ctx.randn((B,N),npdt,seed=2); t=ctx.empty((B,M),npdt)
t0=time.time(); n=0
while time.time()-t0<8:
    for _ in range(20): ctx._call("tramp_matvec_nt", dt, 0, M, N, B, Vt.ptr, x.ptr, t.ptr)
    ctx.sync(); n+=20
el=time.time()-t0
print("dtype",npdt.__name__,"mode",os.environ.get("TRAMP_MFMA_F64_SHAPE"),"B",B,"avg ms",el/n*1e3,"TFLOP/s",2.0*M*N*B*n/el/1e12, flush=True)
PY
PID=$!
sleep 4
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk" | head -8; sleep 1; done
wait $PID
