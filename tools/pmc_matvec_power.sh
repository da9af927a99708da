#!/bin/bash
# Matrix-pipe busy cycles and the clock held (GRBM_GUI_ACTIVE per XCD over the launch) of the f64 dense passes on Gaussian
# and on all-zero operands: the same instruction stream, different toggling.  --pmc only with --kernel-trace.
# Usage (GPU box): tools/pmc_matvec_power.sh -> gpurun_out/r04/pmc_matvec_power/summary.txt
out=$(pwd)/gpurun_out/r04/pmc_matvec_power
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $out/mfma -- python3 tools/pmc_matvec_power.py > $out/mfma.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $out/waits -- python3 tools/pmc_matvec_power.py > $out/waits.txt 2>&1
python3 - $out <<'PY' > $out/summary.txt
import csv, glob, sys
out = sys.argv[1]
def rows(d):
    ct = sorted(glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True))[-1]
    kt = sorted(glob.glob(f"{out}/{d}/**/*kernel_trace.csv", recursive=True))[-1]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))}
    per = {}
    for r in csv.DictReader(open(ct)):
        if "gemm_n" not in r["Kernel_Name"]: continue
        per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = per.get(r["Dispatch_Id"], {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [(dur[k][1], dur[k][2].split("(")[0][:40], dur[k][0], v) for k, v in per.items() if k in dur]
print("f64 dense passes 50000 x 100000 x 64: launches in order - three row-dot, three column on Gaussian operands, then the same on zeros")
a = sorted(rows("mfma")); b = sorted(rows("waits"))
for (t0, name, ns, c), (_, _, _, w) in zip(a, b):
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    ghz = c["GRBM_GUI_ACTIVE"] / 8 / ns
    wc = w["SQ_WAVE_CYCLES"]
    print(f"{name:40s} {ns / 1e6:7.3f} ms  matrix pipe busy {busy:.3f}  clock {ghz:.2f} GHz  busy x clock / 2.4 GHz = {busy * ghz / 2.4:.3f}   "
          f"waves: parked {w['SQ_WAIT_ANY'] / wc:.2f} issue-stalled {w['SQ_WAIT_INST_ANY'] / wc:.2f} lds {w['SQ_WAIT_INST_LDS'] / wc:.3f}")
PY
find $out -name "*.csv" -size +2M -delete
cat $out/summary.txt
