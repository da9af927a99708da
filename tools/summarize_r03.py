"""Turn gpurun_out/prof_r03 (tools/profile_r03.sh) into the committed evidence under profiles/r03/:
kernel_stats of the default bench under rocprofv3, per-kernel PMC tables, traffic.json (HBM bytes per launch
group of the dominant kernels; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, KiB -> B)."""
import collections, csv, glob, json, os, shutil, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r03"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03"
os.makedirs(dst, exist_ok=True)


def pmc(d):
    fs = sorted(glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    if not fs:
        return {}
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return {k: dict(dispatches=len(cnt[k]), **{c: v / len(cnt[k]) for c, v in a.items()}) for k, a in agg.items()}


def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, cut = 0, len(k)
    for i, ch in enumerate(k):                 # the argument list starts at the first "(" outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return k[:cut][:100]


def table(name, dirs):
    per = {d: pmc(d) for d in dirs}
    kernels = sorted({k for p in per.values() for k in p})
    with open(os.path.join(dst, name), "w") as f:
        w = csv.writer(f)
        cols = sorted({c for p in per.values() for v in p.values() for c in v if c != "dispatches"})
        w.writerow(["kernel", "dispatches"] + [c + "_per_dispatch" for c in cols])
        for k in kernels:
            row = {}
            n = 0
            for p in per.values():
                if k in p:
                    n = max(n, p[k]["dispatches"])
                    row.update({c: v for c, v in p[k].items() if c != "dispatches"})
            w.writerow([short(k), n] + ["%.6g" % row[c] if c in row else "" for c in cols])
    return per


ks = sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(dst, "kernel_stats_default_bench_N100000.csv"))
for f, n in (("bench_under_rocprof.json", "bench_under_rocprof_N100000.json"),):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, n))
mat = table("pmc_matrix_kernels_N100000.csv", ["pmc_fetch_mat", "pmc_write_mat", "pmc_mfma_mat", "pmc_waits_a", "pmc_waits_b"])
ew = table("pmc_elementwise_kernels_64x1e6.csv", ["pmc_fetch_ew", "pmc_write_ew", "pmc_valu_ew"])

# traffic.json: launch group = GEMM kernel + fold kernel
R, C, B = 50000, 100000, 64
alg = R * C * 8
ent = {}
f, wr = mat.get("pmc_fetch_mat", {}), mat.get("pmc_write_mat", {})


def grab(d, sub):
    return sum(v.get("FETCH_SIZE", 0) * 2 + v.get("WRITE_SIZE", 0) for k, v in d.items() if sub in k)


for tag, gk in (("matvec_nt", "gemm_nt_kernel<double"), ("matvec_nn", "gemm_nn_kernel<double")):
    kib = grab(f, gk) + grab(wr, gk)
    fold_f = [v for k, v in f.items() if "mfma_fold_kernel<double" in k]
    fold_w = [v for k, v in wr.items() if "mfma_fold_kernel<double" in k]
    kib += (fold_f[0]["FETCH_SIZE"] * 2 if fold_f else 0) + (fold_w[0]["WRITE_SIZE"] if fold_w else 0)
    if kib:
        ent[f"{tag}:f64:R{R}:C{C}:B{B}"] = {"hbm_bytes_per_launch": kib * 1024, "algorithmic_bytes": alg}
for tag, gk in (("matvec_nt", "gemv_nt_kernel<double"), ("matvec_nn", "gemv_nn_kernel<double")):
    kib = grab(f, gk) + grab(wr, gk)
    if kib:
        ent[f"{tag}:f64:R{R}:C{C}:B1"] = {"hbm_bytes_per_launch": kib * 1024, "algorithmic_bytes": alg}
json.dump({"_comment": "HBM bytes per launch group from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of "
                       "tools/pmc_kernels.py 100000 64 on MI355X, round 3). FETCH_SIZE is doubled (gfx950 reports half of a wide "
                       "coalesced stream, MI355X_MICROARCH.md section HBM); counters are KiB. A matvec launch group = GEMM kernel "
                       "+ fold kernel (the fold's figures are the mean over both passes).", "entries": ent},
          open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(ent, indent=1))
# elementwise: bytes moved per component
fe, we, ve = ew.get("pmc_fetch_ew", {}), ew.get("pmc_write_ew", {}), ew.get("pmc_valu_ew", {})
with open(os.path.join(dst, "elementwise_traffic_and_instructions.txt"), "w") as out:
    out.write("per dispatch at 64 x 1e6 components; bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB; T = 8 (f64) / 4 (f32)\n")
    for k in sorted(set(fe) | set(we) | set(ve)):
        if not any(s in k for s in ("ew_", "fft_strided", "fft_block", "spectral")):
            continue
        by = (fe.get(k, {}).get("FETCH_SIZE", 0) * 2 + we.get(k, {}).get("WRITE_SIZE", 0)) * 1024
        T = 4 if "<float" in k else 8
        iv = ve.get(k, {}).get("SQ_INSTS_VALU", 0)
        out.write("%-100s moved %.3f T/component  SQ_INSTS_VALU/component %.1f  (wave instr x64 / 64e6)\n" % (
            short(k), by / (64e6 * T), iv * 64 / 64e6))
print(open(os.path.join(dst, "elementwise_traffic_and_instructions.txt")).read()[:3500])

# fused pair: kernel durations under rocprofv3 --kernel-trace --stats (tools/bench_fused.py f64)
fs = sorted(glob.glob(os.path.join(src, "fused_trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if fs:
    shutil.copy(fs[-1], os.path.join(dst, "kernel_stats_fused_updates_64x1e6.csv"))
if os.path.exists(os.path.join(src, "fused_trace.out")):
    shutil.copy(os.path.join(src, "fused_trace.out"), os.path.join(dst, "fused_updates_under_rocprof.txt"))
if os.path.exists(os.path.join(src, "pmc_summary.txt")):
    shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, "pmc_summary_raw.txt"))
if os.path.exists(os.path.join(src, "kernel_stats_by_grid.csv")):
    shutil.copy(os.path.join(src, "kernel_stats_by_grid.csv"), os.path.join(dst, "kernel_stats_by_grid_default_bench_N100000.csv"))
