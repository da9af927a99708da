#!/bin/bash
# HBM-side traffic (FETCH_SIZE) of the two matrix passes at 128 instances: interleaved sibling chunks vs (tile, chunk) launch
out=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc_chunks
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in 0 1; do
  TRAMP_MFMA_CHUNK_2D=$mode rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/m$mode -- python3 tools/pmc_kernels.py 100000 128 > $out/run$mode.txt 2>&1
  python3 - $out/m$mode $mode <<'PY'
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'gemm_n' not in k: continue
    agg[k[:40]] += float(r['Counter_Value']); cnt[k[:40]].add(r['Dispatch_Id'])
for k in agg:
    gb = agg[k] / len(cnt[k]) * 2 * 1024 / 1e9          # FETCH_SIZE in KiB, doubled per the gfx950 correction
    print(f"TRAMP_MFMA_CHUNK_2D={sys.argv[2]} {k:40s} fetched {gb:7.1f} GB per launch = {gb/40.0:4.2f} x the 40 GB matrix")
PY
done
find $out -name "*.csv" -size +2M -delete
