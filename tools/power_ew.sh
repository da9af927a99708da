#!/bin/bash
# rocm-smi power / clock samples while one elementwise update loops at 64 x 1e6 (f64).
# Usage: tools/power_ew.sh > profiles/r02/power_log_elementwise.txt
for OP in relu_bwd sgn_lik abs_lik gb_prior; do
python3 - $OP <<'PY' &
import sys, time; sys.path.insert(0,'.')
import numpy as np, tramp_amd as T
op=sys.argv[1]; B=64; N=1_000_000
ctx=T.default_context()
bz=ctx.randn((B,N),np.float64,seed=1,scale=1.5); bx=ctx.randn((B,N),np.float64,seed=2,scale=1.5)
y=ctx.array(np.sign(np.random.RandomState(0).standard_normal((B,N))))
az=ctx.scalars(1.3,B); ax=ctx.scalars(0.7,B); a_new=ctx.empty((B,)); b_new=ctx.empty((B,N))
kw=dict(out_a=a_new,out_b=b_new)
f={"relu_bwd": (lambda c=T.ReluChannel(ctx=ctx): c.compute_backward_message(az,bz,ax,bx,**kw)),
   "sgn_lik": (lambda l=T.SgnLikelihood(y=y,ctx=ctx): l.compute_backward_message(az,bz,**kw)),
   "abs_lik": (lambda l=T.AbsLikelihood(y=y,ctx=ctx): l.compute_backward_message(az,bz,**kw)),
   "gb_prior": (lambda p=T.GaussBernoulliPrior(size=N,rho=0.05,ctx=ctx): p.compute_forward_message(ax,bx,**kw))}[op]
f(); ctx.sync()
t0=time.time(); n=0
while time.time()-t0<8:
    for _ in range(50): f()
    ctx.sync(); n+=50
el=time.time()-t0
print(f"[workload] {op} 64 x 1e6 f64: {el/n*1e6:.1f} us per update", flush=True)
PY
PID=$!
sleep 7
for i in $(seq 1 6); do
  echo "[sample $OP t=$i] $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Current Socket Graphics Package Power|sclk clock level' | sed -E 's/^GPU\[0\]\s*:\s*//' | tr '\n' ';')"
  sleep 0.5
done
wait $PID
done
