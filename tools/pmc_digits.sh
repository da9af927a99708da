#!/bin/bash
# rocprofv3 passes over the int8-digit dense pass (csrc/gemm_sliced.h) at the headline shape: HBM traffic (FETCH_SIZE
# doubled on gfx950, see MI355X_MICROARCH.md), matrix-pipe busy cycles, where the waves wait.  --pmc only with --kernel-trace.
# Usage (GPU box): tools/pmc_digits.sh  ->  gpurun_out/r04/pmc_digits/summary.txt
out=$(pwd)/gpurun_out/r04/pmc_digits
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/pmc_digits.py > $out/fetch.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/pmc_digits.py > $out/write.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $out/mfma -- python3 tools/pmc_digits.py > $out/mfma.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $out/waits -- python3 tools/pmc_digits.py > $out/waits.txt 2>&1
python3 - $out <<'PY' > $out/summary.txt
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set)); dur = collections.defaultdict(list)
for d in ("fetch", "write", "mfma", "waits"):
    fs = sorted(glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name'].split('(')[0][:60]
        if not any(s in k for s in ("gemm_sliced", "sl_", "mfma_fold")): continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
    fs = sorted(glob.glob(f"{out}/{d}/**/*kernel_trace.csv", recursive=True))
    if d == "fetch" and fs:
        for r in csv.DictReader(open(fs[-1])):
            k = r['Kernel_Name'].split('(')[0][:60]
            if any(s in k for s in ("gemm_sliced", "sl_", "mfma_fold")): dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print("int8-digit dense pass, 50000 x 100000 x 64, 7 digits: per launch (rocprofv3 --pmc, one counter group per run; FETCH_SIZE doubled per the gfx950 correction)")
for k, c in agg.items():
    per = {n: v / len(cnt[k][n]) for n, v in c.items()}
    line = f"{k:60s}"
    if dur[k]: line += f" {sum(dur[k]) / len(dur[k]) / 1e3:9.1f} us"
    if "FETCH_SIZE" in per: line += f"  fetch {2 * per['FETCH_SIZE'] * 1024 / 1e9:7.2f} GB"
    if "WRITE_SIZE" in per: line += f"  write {per['WRITE_SIZE'] * 1024 / 1e9:6.2f} GB"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per and per.get("GRBM_GUI_ACTIVE"):
        line += f"  matrix pipe busy {per['SQ_VALU_MFMA_BUSY_CYCLES'] / (per['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}  cycles/XCD {per['GRBM_GUI_ACTIVE'] / 8:.3e}"
        if dur[k]: line += f" = {per['GRBM_GUI_ACTIVE'] / 8 / (sum(dur[k]) / len(dur[k])):.2f} GHz"
    if per.get("SQ_WAVE_CYCLES"):
        w = per["SQ_WAVE_CYCLES"]
        line += f"  waves: parked {per.get('SQ_WAIT_ANY', 0) / w:.2f} issue-stalled {per.get('SQ_WAIT_INST_ANY', 0) / w:.2f} lds-wait {per.get('SQ_WAIT_INST_LDS', 0) / w:.3f} active {per.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f}"
    print(line)
PY
find $out -name "*.csv" -size +2M -delete
cat $out/summary.txt
