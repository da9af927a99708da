#!/bin/bash
# kernels per EP iteration at the reference's benchmark size (N = 1000, one instance)
out=$GRAFT_REPO_ROOT/gpurun_out/r02/smalltrace
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/small_one.py <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import tramp_amd as T
ctx = T.default_context()
N, B = 1000, 1
M, rho, nv = N // 2, 0.05, 1e-2
rs = np.random.RandomState(1)
W = rs.randn(M, N) / np.sqrt(N)
X = rs.standard_normal((B, N)) * rs.binomial(1, rho, (B, N))
Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((B, M))
model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x")
         @ T.LinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z")
         @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
ep = T.ExpectationPropagation(model, ctx=ctx)
ep.iterate(max_iter=5, callback=T.PassCallback(), damping=0.1)
ep.iterate(max_iter=200, callback=T.PassCallback(), damping=0.1, check_every=1 << 30, warm_start=True)
ctx.sync()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /tmp/small_one.py > $out/run.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r02/smalltrace/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print(r["Calls"], f'{float(r["AverageNs"])/1e3:7.1f} us', r["Name"][:110])
PY
find $out -name "*kernel_trace.csv" -size +2M -delete
