#!/bin/bash
# Where the waves of the two matrix passes spend their cycles (batch 64): wait / active counters, two passes of 4
out=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc_stalls
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $out/a -- python3 tools/pmc_kernels.py 100000 64 > $out/a.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/b -- python3 tools/pmc_kernels.py 100000 64 > $out/b.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/c -- python3 tools/pmc_kernels.py 100000 64 > $out/c.txt 2>&1
python3 - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for d in "abc":
    fs = sorted(glob.glob(f"gpurun_out/r02/pmc_stalls/{d}/**/*counter_collection.csv", recursive=True))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name']
        if 'gemm_n' not in k: continue
        agg[k[:34]][r['Counter_Name']] += float(r['Counter_Value']); cnt[k[:34]][r['Counter_Name']].add(r['Dispatch_Id'])
for k, c in agg.items():
    print(k, {n: f"{v/len(cnt[k][n]):.3e}" for n, v in sorted(c.items())})
PY
find $out -name "*.csv" -size +2M -delete
