"""Turn gpurun_out/prof_r04 (tools/profile_r04.sh) into the committed evidence under profiles/r04/: kernel_stats of the
default bench under rocprofv3, per-kernel PMC tables of the componentwise updates (the resident single-pass kernels and
K7's tile kernels) and a readable summary: HBM traffic per launch in units of T = one 64 x 1e6 array (FETCH_SIZE doubled
per the gfx950 correction of MI355X_MICROARCH.md, KB -> B), VALU instructions per component, wave-cycle breakdown."""
import collections, csv, glob, json, os, shutil, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r04"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04"
os.makedirs(dst, exist_ok=True)
NCOMP = 64e6


def pmc(d):
    fs = sorted(glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    if not fs:
        return {}
    for r in csv.DictReader(open(fs[-1])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return {k: dict(dispatches=len(cnt[k]), **{c: v / len(cnt[k]) for c, v in a.items()}) for k, a in agg.items()}


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


ks = sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if ks:
    with open(ks[-1]) as f, open(os.path.join(dst, "kernel_stats_default_bench_N100000.csv"), "w") as g:
        rd = csv.reader(f)
        w = csv.writer(g)
        for row in rd:                                         # rocBLAS / Tensile names are 2 KB each: shorten
            row[0] = row[0][:140]
            w.writerow(row)
for f, n in (("bench_under_rocprof.json", "bench_under_rocprof_N100000.json"), ("kernel_stats_by_grid.csv", "kernel_stats_by_grid_default_bench_N100000.csv"),
             ("kernel_durations_ew.csv", "kernel_durations_elementwise_64x1e6.csv"), ("pmc_summary.txt", "pmc_summary_raw.txt")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, n))

dur = {}
p = os.path.join(src, "kernel_durations_ew.csv")
if os.path.exists(p):
    for r in csv.DictReader(open(p)):
        dur[short(r["kernel"])] = float(r["avg_ns"]) / 1e3
fetch, write, valu, waits = pmc("pmc_fetch_ew"), pmc("pmc_write_ew"), pmc("pmc_valu_ew"), pmc("pmc_waits_ew")
k7a, k7b = pmc("pmc_k7_a"), pmc("pmc_k7_b")
with open(os.path.join(dst, "elementwise_traffic_and_instructions.txt"), "w") as f:
    f.write("Componentwise factor updates at 64 x 1e6 (tools/pmc_ew.py under rocprofv3 --pmc, one counter set per pass;\n"
            "tools/profile_r04.sh).  T = 512 MB (f64) / 256 MB (f32) = one array of the batch.  fetch = FETCH_SIZE x 2 (gfx950:\n"
            "the counter tallies 128-byte requests at 64 B, MI355X_MICROARCH.md), write = WRITE_SIZE, both KB -> bytes.\n"
            "VALU/comp = SQ_INSTS_VALU x 64 / 64e6.  parked / stalled / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY /\n"
            "SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.\n\n")
    f.write(f"{'kernel':100s} {'us':>8s} {'fetch/T':>8s} {'write/T':>8s} {'moved/T':>8s} {'VALU/comp':>10s} {'LDS/comp':>9s} {'parked':>7s} {'stalled':>8s} {'active':>7s}\n")
    names = sorted(set(fetch) | set(write), key=lambda k: -dur.get(k, 0))
    for k in names:
        if not any(s in k for s in ("ew_resident", "ew_moments_tile", "ew_apply_tile", "var_post", "fft_strided_kernel")):
            continue
        T = 512e6 if ("double" in k) else 256e6
        fe = fetch.get(k, {}).get("FETCH_SIZE", float("nan")) * 2 * 1e3 / T
        wr = write.get(k, {}).get("WRITE_SIZE", float("nan")) * 1e3 / T
        v = valu.get(k, {})
        wt = waits.get(k, {})
        wc = wt.get("SQ_WAVE_CYCLES", float("nan"))
        f.write(f"{k:100s} {dur.get(k, float('nan')):8.1f} {fe:8.3f} {wr:8.3f} {fe + wr:8.3f} "
                f"{v.get('SQ_INSTS_VALU', float('nan')) * 64 / NCOMP:10.1f} {v.get('SQ_INSTS_LDS', float('nan')) * 64 / NCOMP:9.2f} "
                f"{wt.get('SQ_WAIT_ANY', float('nan')) / wc:7.2f} {wt.get('SQ_WAIT_INST_ANY', float('nan')) / wc:8.2f} {wt.get('SQ_ACTIVE_INST_ANY', float('nan')) / wc:7.2f}\n")
    f.write("\nK7 (two-region channels, tile kernels), the cycle-level counters (pmc_k7_a / pmc_k7_b passes):\n")
    f.write(f"{'kernel':100s} {'VALU/comp':>10s} {'LDS/comp':>9s} {'active_valu/wave_cyc':>21s} {'wait_lds/wave_cyc':>18s} {'parked':>7s} {'stalled':>8s}\n")
    for k in sorted(k7a):
        if "ew_moments_tile" not in k and "ew_apply_tile" not in k:
            continue
        a, b = k7a[k], k7b.get(k, {})
        wc = b.get("SQ_WAVE_CYCLES", float("nan"))
        f.write(f"{k:100s} {a.get('SQ_INSTS_VALU', float('nan')) * 64 / NCOMP:10.1f} {a.get('SQ_INSTS_LDS', float('nan')) * 64 / NCOMP:9.2f} "
                f"{a.get('SQ_ACTIVE_INST_VALU', float('nan')) / wc:21.3f} {a.get('SQ_WAIT_INST_LDS', float('nan')) / wc:18.3f} "
                f"{b.get('SQ_WAIT_ANY', float('nan')) / wc:7.2f} {b.get('SQ_WAIT_INST_ANY', float('nan')) / wc:8.2f}\n")
    f.write("\n(SQ_INSTS_VALU_TRANS and SQ_INST_CYCLES_VALU, asked for by VERDICT r3, are not exposed by rocprofv3 on gfx950 / ROCm 7.2:\n"
            "the passes that name them return the other counters only.)\n")
print(open(os.path.join(dst, "elementwise_traffic_and_instructions.txt")).read())
