#!/bin/bash
# HBM traffic per launch of the dominant dense kernels at the headline shape (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE
# in separate runs, --kernel-trace only) -> profiles/r04/traffic.json, which bench.py quotes as roofline.traffic.
# Usage (GPU box): tools/traffic_r04.sh
out=$(pwd)/gpurun_out/r04/traffic
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/pmc_kernels.py 100000 64 > $out/fetch.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/pmc_kernels.py 100000 64 > $out/write.txt 2>&1
python3 - $out <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
def pmc(d, counter):
    fs = sorted(glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True))
    agg, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(fs[-1])):
        if r["Counter_Name"] != counter: continue
        agg[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: v / len(cnt[k]) for k, v in agg.items()}
f, w = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
R, C = 50000, 100000
ent = {}
def kib(sub, grid_filter=None):
    return sum(2 * v for k, v in f.items() if sub in k), sum(v for k, v in w.items() if sub in k)
fold = kib("mfma_fold_kernel<double")
for tag, gk, B in (("matvec_nt", "gemm_nt_kernel<double", 64), ("matvec_nn", "gemm_nn_kernel<double", 64),
                   ("matvec_nt", "gemv_nt_kernel<double", 1), ("matvec_nn", "gemv_nn_kernel<double", 1)):
    a, b = kib(gk)
    if not a: continue
    tot = a + b + ((fold[0] + fold[1]) if B == 64 else 0)
    ent[f"{tag}:f64:R{R}:C{C}:B{B}"] = {"hbm_bytes_per_launch": tot * 1024, "algorithmic_bytes": R * C * 8,
                                         "fetch_bytes": a * 1024, "write_bytes": b * 1024}
json.dump({"_comment": "HBM bytes per launch group from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of "
                       "tools/pmc_kernels.py 100000 64 on MI355X, round 4, the final library). FETCH_SIZE is doubled (gfx950 reports "
                       "half of a wide coalesced stream, MI355X_MICROARCH.md section HBM); counters are KiB. A batch-64 launch "
                       "group = GEMM kernel + fold kernel.", "entries": ent}, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(ent, indent=1))
PY
find $out -name "*.csv" -size +2M -delete
