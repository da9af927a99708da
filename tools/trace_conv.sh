#!/bin/bash
# per-kernel durations of the structured (ConvChannel) EP iteration at 64 x 1e6
out=$GRAFT_REPO_ROOT/gpurun_out/r02/convtrace
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_conv.py 1000000 64 f64 > $out/bench.txt 2>&1
tail -2 $out/bench.txt
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r02/convtrace/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Calls"], f'{float(r["AverageNs"])/1e3:9.1f} us', r["Percentage"], r["Name"][:100])
PY
find $out -name "*kernel_trace.csv" -size +2M -delete
