#!/bin/bash
# VALU instruction counts and busy cycles of the elementwise kernels at 64 x 1e6 (one --pmc pass)
out=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc_valu_ew
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out -- python3 tools/pmc_ew.py > $out/run.txt 2>&1
python3 - <<PY
import collections, csv, glob
f = sorted(glob.glob("gpurun_out/r02/pmc_valu_ew/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'double' not in k or not ('moments' in k or 'apply' in k): continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k, c in sorted(agg.items()):
    n = len(cnt[k])
    v = c['SQ_INSTS_VALU'] / n
    print(f"{k[:75]:75s} VALU/comp {v*64/64e6:6.1f}  LDS/comp {c['SQ_INSTS_LDS']/n*64/64e6:5.1f}  gui_cycles/XCD {c['GRBM_GUI_ACTIVE']/n/8:9.0f}  active_valu {c['SQ_ACTIVE_INST_VALU']/n:.3g}")
PY
find $out -name "*.csv" -size +2M -delete
