#!/bin/bash
# MFMA pipe utilisation of the two matrix passes at 64 and 128 instances (interleaved sibling chunks)
out=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc_mfma_chunks
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for b in 64 128; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $out/b$b -- python3 tools/pmc_kernels.py 100000 $b > $out/run$b.txt 2>&1
  python3 - $out/b$b $b <<'PY'
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'gemm_n' not in k: continue
    agg[k[:40]][r['Counter_Name']] += float(r['Counter_Value']); cnt[k[:40]].add(r['Dispatch_Id'])
for k, c in agg.items():
    n = len(cnt[k])
    util = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)
    print(f"batch {sys.argv[2]:>3s} {k:40s} MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) = {util:5.3f}   cycles per XCD {c['GRBM_GUI_ACTIVE']/n/8:.3e}")
PY
done
find $out -name "*.csv" -size +2M -delete
