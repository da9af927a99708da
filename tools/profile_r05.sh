#!/bin/bash
# Round-5 profiles (run on the GPU box through gpurun): rocprofv3 kernel-trace stats of the DEFAULT bench command and
# separate --pmc passes over the componentwise updates (the resident single-pass kernels of csrc/ew_resident.h and the
# tile kernels of K7).  Never --pmc together with a trace domain other than --kernel-trace.
# Usage: tools/profile_r05.sh [stats|pmc_ew|pmc_k7|all]
set -u
WHAT=${1:-all}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  # rocprofv3 (ROCm 7.2) segfaults inside rocSOLVER's syevd for k >= 24576: TRAMP_EIG_CAP=16384 keeps every
  # library eigensolve below that (the spectral split simply recurses once more); everything else is the default.
  TRAMP_EIG_CAP=16384 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
  tail -3 $OUT/stats.err
fi
cd $REPO
if [ "$WHAT" = pmc_ew ] || [ "$WHAT" = all ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_ew -- python3 tools/pmc_ew.py > $OUT/fetch_ew.out 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_ew -- python3 tools/pmc_ew.py > $OUT/write_ew.out 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_valu_ew -- python3 tools/pmc_ew.py > $OUT/valu_ew.out 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_waits_ew -- python3 tools/pmc_ew.py > $OUT/waits_ew.out 2>&1
fi
if [ "$WHAT" = pmc_k7 ] || [ "$WHAT" = all ]; then
  # the cycle-level counters VERDICT r3 asked for on K7 (two-region channels): transcendental instructions, VALU
  # issue cycles, LDS issue stalls - per component in the summary
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc_k7_a -- python3 tools/pmc_ew.py 64 1000000 k7 > $OUT/k7_a.out 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_k7_b -- python3 tools/pmc_ew.py 64 1000000 k7 > $OUT/k7_b.out 2>&1
fi
for d in $OUT/pmc_*; do echo "== $d"; python3 tools/pmc_parse.py $d 2>/dev/null | cut -c1-700; done > $OUT/pmc_summary.txt 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
fs = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if fs:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        name = r["Kernel_Name"]
        if not any(k in name for k in ("gemm_n", "gemm_sliced", "sl_", "mfma_fold", "gemv_n", "nn_fold", "ew_", "var_post", "lin_", "ep_", "damp")):
            continue
        grid = "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        agg[(name.split("(")[0][:110], grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, "kernel_stats_by_grid.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "calls", "avg_ns", "min_ns", "max_ns", "total_ns"])
        for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, g, len(v), sum(v) / len(v), min(v), max(v), sum(v)])
# kernel durations of the pmc_ew script (kernel trace of the FETCH pass)
for d, nm in (("pmc_fetch_ew", "kernel_durations_ew.csv"),):
    fs = sorted(glob.glob(os.path.join(out, d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        agg[r["Kernel_Name"].split("(")[0][:120]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, nm), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "avg_ns", "min_ns"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v) / len(v), min(v)])
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
