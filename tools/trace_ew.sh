#!/bin/bash
# per-kernel durations of the elementwise updates at 64 x 1e6 (rocprofv3 --kernel-trace --stats)
out=$GRAFT_REPO_ROOT/gpurun_out/r02/ewtrace
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_ew.py > $out/bench.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r02/ewtrace/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "double" in n and ("moments" in n or "apply" in n): print(r["Calls"], r["AverageNs"], n[:110])
PY
find $out -name "*kernel_trace.csv" -size +2M -delete
