"""gpurun_out/prof_r06 (tools/profile_r06.sh) -> the evidence committed under profiles/r06/: kernel stats of the default bench
under rocprofv3 (whole and by grid), and per-kernel PMC tables of the componentwise updates - HBM bytes per launch in units
of T = one 64 x 1e6 array (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, KB -> B), VALU instructions
per component, wave-cycle breakdown.  python tools/summarize_r06.py <src> <dst>"""
import collections
import csv
import glob
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r06"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r06"
os.makedirs(dst, exist_ok=True)
NCOMP = 64e6


def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, cut = 0, len(k)
    for i, ch in enumerate(k):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return k[:cut][:110]


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def pmc(d):
    f = newest(os.path.join(d, "**", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return {k: dict(dispatches=len(cnt[k]), **{c: v / len(cnt[k]) for c, v in a.items()}) for k, a in agg.items()}


def durations(d):
    f = newest(os.path.join(d, "**", "*kernel_trace.csv"))
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: (sum(v) / len(v) / 1e3, len(v)) for k, v in agg.items()}


# ---- default bench: kernel stats, whole and by grid
ks = newest(os.path.join("stats", "**", "*kernel_stats.csv"))
if ks:
    with open(ks) as f, open(os.path.join(dst, "kernel_stats_default_bench_N100000.csv"), "w") as g:
        w = csv.writer(g)
        for row in csv.reader(f):
            row[0] = row[0][:140]                   # rocBLAS / Tensile names are 2 KB each
            w.writerow(row)
kt = newest(os.path.join("stats", "**", "*kernel_trace.csv"))
if kt:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        name = r["Kernel_Name"]
        if not any(k in name for k in ("gemm_n", "gemm_sliced", "sl_", "mfma_fold", "gemv_n", "nn_fold", "ew_", "var_post", "lin_", "ep_", "damp")):
            continue
        grid = "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        agg[(short(name), grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(dst, "kernel_stats_by_grid_default_bench_N100000.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "calls", "avg_ns", "min_ns", "max_ns", "total_ns"])
        for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, g, len(v), sum(v) / len(v), min(v), max(v), sum(v)])
if os.path.exists(os.path.join(src, "bench_under_rocprof.json")):
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "bench_under_rocprof_N100000.json"))

# ---- componentwise updates under the counters
for script, title, out in (("pmc_ew_r6", "the updates added in round 6 (tools/pmc_ew_r6.py)", "elementwise_traffic_and_instructions_round6_factors.txt"),
                           ("pmc_ew", "every componentwise update of rounds 1-5 (tools/pmc_ew.py)", "elementwise_traffic_and_instructions.txt")):
    fetch, write, valu, waits = (pmc(f"{script}_{p}") for p in ("fetch", "write", "valu", "waits"))
    dur = durations(f"{script}_fetch")
    if not fetch and not write:
        continue
    with open(os.path.join(dst, out), "w") as f:
        f.write(f"Componentwise factor updates at 64 x 1e6, {title} under rocprofv3 --pmc, one counter set per pass\n"
                "(tools/profile_r06.sh).  T = 512 MB (f64) / 256 MB (f32) = one array of the batch.  fetch = FETCH_SIZE x 2 (gfx950: the\n"
                "counter tallies 128-byte requests at 64 B, MI355X_MICROARCH.md), write = WRITE_SIZE, both KB -> bytes.  VALU/comp =\n"
                "SQ_INSTS_VALU x 64 / 64e6 (modulus likelihood: 32e6 complex components - halve its figure per real coordinate).\n"
                "parked / stalled / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.  us = the kernel's\n"
                "average duration in the FETCH pass's kernel trace (a cold-start process, counters attached).\n\n")
        f.write(f"{'kernel':100s} {'calls':>5s} {'us':>8s} {'fetch/T':>8s} {'write/T':>8s} {'moved/T':>8s} {'VALU/comp':>10s} {'LDS/comp':>9s} {'parked':>7s} {'stalled':>8s} {'active':>7s}\n")
        for k in sorted(set(fetch) | set(write), key=lambda k: -dur.get(k, (0, 0))[0]):
            if not any(s in k for s in ("ew_resident", "ew_moments_tile", "ew_apply_tile", "modulus_", "ew_res_prepare", "ew_finalize")):
                continue
            T = 512e6 if "double" in k else 256e6
            fe = fetch.get(k, {}).get("FETCH_SIZE", float("nan")) * 2 * 1e3 / T
            wr = write.get(k, {}).get("WRITE_SIZE", float("nan")) * 1e3 / T
            v, wt = valu.get(k, {}), waits.get(k, {})
            wc = wt.get("SQ_WAVE_CYCLES", float("nan"))
            d, n = dur.get(k, (float("nan"), 0))
            f.write(f"{k:100s} {n:5d} {d:8.1f} {fe:8.3f} {wr:8.3f} {fe + wr:8.3f} "
                    f"{v.get('SQ_INSTS_VALU', float('nan')) * 64 / NCOMP:10.1f} {v.get('SQ_INSTS_LDS', float('nan')) * 64 / NCOMP:9.2f} "
                    f"{wt.get('SQ_WAIT_ANY', float('nan')) / wc:7.2f} {wt.get('SQ_WAIT_INST_ANY', float('nan')) / wc:8.2f} {wt.get('SQ_ACTIVE_INST_ANY', float('nan')) / wc:7.2f}\n")
    print(open(os.path.join(dst, out)).read()[:6000])
