#!/usr/bin/env python3
"""Headline benchmark: EP iterations/s x N x instances (msg-components/s) for the sparse GLM
(GaussBernoulli prior + dense LinearChannel + Gaussian likelihood, Gaussian ensemble,
alpha = 0.5, fp64) on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full EP iteration (forward sweep, backward sweep, variable posteriors) over
the rank's batch of independent problem instances that share the sensing matrix W (the
ensemble axis of BASELINE.json configs[3]).  Instances are sharded across ranks with no
data-path collective: rank 0 draws W and broadcasts it once over RCCL (setup, untimed), every
rank factorises it and runs its own instances ("weak" scaling: per-GPU work fixed).
Setup (W generation, broadcast, factorisation, teacher data) is reported separately, as the
reference reports svd_time separately (examples/figures/compute_benchmark.py:21-27).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}   # fp64: AMD datasheet (matrix = vector rate); fp32: guide
# measured on the box by tools/probes/probe_peaks.hip (back-to-back independent MFMAs, 16 waves/CU)
MFMA_MEASURED_TFLOPS = {"f64": 47.5, "f32": 146.3}

TAGS = ["ew_factor_update", "matvec_nt", "matvec_nn", "linear_scalars", "variable_posterior", "other", "fft"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", dest="n", type=int, default=int(os.environ.get("TRAMP_BENCH_N", 65536)),
                    help="signal dimension N (not --n: torch.distributed.run would read it as its own option)")
    ap.add_argument("--alpha", type=float, default=0.5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("TRAMP_BENCH_BATCH", 64)),
                    help="instances per GPU")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--rho", type=float, default=0.05)
    ap.add_argument("--noise-var", type=float, default=1e-2)
    ap.add_argument("--damping", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-denoiser", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=4000)
    ap.add_argument("--save-factors", default=None, metavar="DIR",
                    help="after the setup, dump the thin factors (Uk, Vt, sing) and the observations to DIR")
    ap.add_argument("--load-factors", default=None, metavar="DIR",
                    help="skip W generation / factorisation and load what --save-factors wrote (used to put the "
                         "timed region under rocprofv3, which crashes inside rocSOLVER's syevd at k >= 24576)")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(args):
    """The oracle (NumPy restatement of the reference's path, reference formulation with the
    full SVD) timed on the host cores, on a bounded sample of the same workload."""
    from oracle import tramp_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    N = args.cpu_n
    M = int(args.alpha * N)
    rs = np.random.RandomState(1)
    W = orc.gaussian_ensemble(rs, M, N)
    x = orc.gb_sample(rs, N, args.rho)
    y = W @ x + np.sqrt(args.noise_var) * rs.standard_normal(M)
    t0 = time.perf_counter()
    lin = orc.LinearFactor(W)
    setup_s = time.perf_counter() - t0
    fs = [orc.GBPriorFactor(N, rho=args.rho), lin, orc.GaussLikFactor(y, args.noise_var)]
    iters = 300
    orc.ep_chain(fs, [N, M], max_iter=2, damping=args.damping, tol=0.0)       # warm-up

    def run(n_it):
        t0 = time.perf_counter()
        res = orc.ep_chain(fs, [N, M], max_iter=n_it, damping=args.damping, tol=0.0, max_increase=np.inf)
        return res["n_iter"], time.perf_counter() - t0
    runs = [run(iters) for _ in range(5)]                                     # median of >= 5 runs
    n_it, dt = sorted(runs, key=lambda r: r[1])[len(runs) // 2]
    out = {"value": N * n_it / dt, "unit": "msg-components/s", "cores": int(threads), "kind": "port",
           "sample": f"oracle EP (reference formulation: full SVD, 4 dense mat-vecs per solve), 1 instance, "
                     f"N={N}, alpha={args.alpha}, median of 5 runs of {n_it} iterations ({dt:.2f} s each; "
                     f"+{setup_s:.1f} s SVD setup, untimed), NumPy/OpenBLAS {threads} threads, "
                     f"{os.cpu_count()} host cores ({_cpu_model()})"}
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            n1, dt1 = run(60)
        out["single_thread_value"] = N * n1 / dt1
    except Exception:                                                          # noqa: BLE001
        pass
    return out


def denoiser_roofline(T, ctx, dtype):
    """Gauss-Bernoulli denoiser (K1) at the BASELINE vector size N = 1e6: single instance
    (latency regime, data stays in cache) and a 64-instance ensemble (512 MB per array, past
    the 256 MB Infinity Cache); plus 4096 instances of n = 4096, where an instance fits one
    workgroup's registers and the update is a single pass.  Algorithmic traffic 2T bytes per component."""
    out = {}
    np_dt = np.float64 if dtype == "f64" else np.float32
    for B, N in ((1, 1_000_000), (64, 1_000_000), (4096, 4096)):
        bx = ctx.randn((B, N), np_dt, seed=3, scale=1.5)
        ax = ctx.scalars(1.3, B)
        p = T.GaussBernoulliPrior(size=N, rho=0.05, dtype=np_dt, ctx=ctx)
        a_new = ctx.empty((B,), np.float64)
        b_new = ctx.empty((B, N), np_dt)
        for _ in range(3):
            p.compute_forward_message(ax, bx, out_a=a_new, out_b=b_new)
        reps = 50 if B == 1 else 10
        e0, e1 = ctx.event(), ctx.event()
        e0.record()
        for _ in range(reps):
            p.compute_forward_message(ax, bx, out_a=a_new, out_b=b_new)
        e1.record()
        ms = e0.elapsed_ms(e1) / reps
        alg = 2 * np.dtype(np_dt).itemsize * N * B
        out[f"B{B}" if N == 1_000_000 else f"B{B}_n{N}"] = {"n": N, "batch": B, "us_per_update": ms * 1e3, "algorithmic_GBps": alg / ms / 1e6,
                        "frac_of_hbm_peak": alg / ms / 1e6 / HBM_PEAK_GBS,
                        "components_per_sec": N * B / (ms * 1e-3)}
        del bx, b_new
    return out


def structured_n1e6(T, ctx, dtype):
    """EP at BASELINE's literal vector size N = 10^6 with a matrix-free operator (dense W cannot exist
    there): sparse deconvolution GaussBernoulli -> ConvChannel(dense random circulant filter) -> Gaussian
    likelihood, 64 instances, on the hand-written two-pass FFT.  An auxiliary figure, never `value`."""
    np_dt = np.float64 if dtype == "f64" else np.float32
    n, B, rho, nv = 1_000_000, 64, 0.05, 1e-3
    rs = np.random.RandomState(3)
    ch = T.ConvChannel(rs.randn(n) / np.sqrt(n), dtype=np_dt, ctx=ctx)
    X = (rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n))).astype(np_dt)
    Y = ch.convolve(ctx.array(X, np_dt)).get() + np.sqrt(nv) * rs.standard_normal((B, n)).astype(np_dt)
    model = (T.GaussBernoulliPrior(size=n, rho=rho, dtype=np_dt, ctx=ctx) @ T.SISOVariable(id="x") @ ch
             @ T.SISOVariable(id="z") @ T.GaussianLikelihood(y=Y, var=nv, dtype=np_dt, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    never = 1 << 30
    ep.iterate(max_iter=3, callback=T.PassCallback(), damping=0.1, check_every=never)
    ctx.sync()
    K = 20
    t0 = time.perf_counter()
    ep.iterate(max_iter=K, callback=T.PassCallback(), damping=0.1, warm_start=True, check_every=never)
    ctx.sync()
    dt = time.perf_counter() - t0
    return {"workload": "sparse deconvolution EP, ConvChannel (circulant, matrix-free), damping 0.1", "n": n, "instances": B,
            "ms_per_iteration": dt / K * 1e3, "msg_components_per_sec": n * B * K / dt,
            "mse": float(np.mean(ep.mse("x", X)))}


class _Shape:
    "Stand-in for W when the factors are supplied: LinearChannel only reads .shape then."

    def __init__(self, m, n):
        self.shape = (m, n)


def run(args):
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks",
                  file=sys.stderr)
            sys.exit(2)
    dist = None
    torch = None
    # TRAMP_BENCH_FORCE_DIST=1 takes the torch.distributed / RCCL path even with one rank
    # (used to smoke-test the multi-GPU code on a 1-GPU box)
    if world > 1 or os.environ.get("TRAMP_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import tramp_amd as T
    from tramp_amd import ensemble
    ctx = T.Context(local_rank)
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    N, B = args.n, args.batch
    M = int(args.alpha * N)

    def barrier():
        ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    # ---------------- setup (untimed, reported)
    t_setup = time.perf_counter()
    bcast_s = None
    start, stop = ensemble.shard_range(B * world, rank, world)
    if args.load_factors:
        d = args.load_factors
        k = min(M, N)
        Uk = ctx.array(np.load(os.path.join(d, "Uk.npy"), mmap_mode="r"), np_dt)
        Vt = ctx.array(np.load(os.path.join(d, "Vt.npy"), mmap_mode="r"), np_dt)
        sing = np.load(os.path.join(d, "sing.npy"))
        X = np.load(os.path.join(d, "X.npy"))
        Y = np.load(os.path.join(d, "Y.npy"))
        assert Uk.shape == (M, k) and Vt.shape == (k, N) and X.shape == (B, N), "factors do not match --size/--batch"
        tf = time.perf_counter()
        lin = T.LinearChannel(_Shape(M, N), dtype=np_dt, ctx=ctx,
                              factors=(Uk, sing, Vt, int(np.sum(sing > 0))))
        factorize_s = 0.0
    else:
        if dist is None:
            W = ctx.randn((M, N), np_dt, seed=1234, scale=1.0 / np.sqrt(N))
        else:
            # shared sensing matrix: rank 0 draws it, RCCL broadcast over xGMI to the other ranks
            Wt = torch.empty((M, N), dtype=torch.float64 if args.dtype == "f64" else torch.float32, device="cuda")
            W = T.DeviceArray(ctx, (M, N), np_dt, ptr=Wt.data_ptr(), owner=False)
            if rank == 0:
                ctx._call("tramp_randn", T.device.dtype_code(np_dt), W.ptr, W.size, 1234, 1.0 / np.sqrt(N))
                ctx.sync()
            bcast_s = ensemble.broadcast_tensor(Wt, src=0)
        # teacher data of this rank's instances: block [start, stop) of the B*world ensemble
        rs = np.random.RandomState(ensemble.instance_seeds(1000, start, stop)[0] % (2**31))
        X = (rs.standard_normal((B, N)) * rs.binomial(1, args.rho, (B, N))).astype(np_dt)
        Xd = ctx.array(X, np_dt)
        Zd = ctx.empty((B, M), np_dt)
        ctx._call("tramp_matvec_nt", T.device.dtype_code(np_dt), 0, M, N, B, W.ptr, Xd.ptr, Zd.ptr)
        Y = Zd.get() + np.sqrt(args.noise_var) * rs.standard_normal((B, M))
        tf = time.perf_counter()
        lin = T.LinearChannel(W, dtype=np_dt, ctx=ctx, svd="device")
        ctx.sync()
        factorize_s = time.perf_counter() - tf
        del W
        if dist is not None:
            del Wt
            torch.cuda.empty_cache()
        if args.save_factors and rank == 0:
            os.makedirs(args.save_factors, exist_ok=True)
            np.save(os.path.join(args.save_factors, "Uk.npy"), lin._Uk.get())
            np.save(os.path.join(args.save_factors, "Vt.npy"), lin._Vt.get())
            np.save(os.path.join(args.save_factors, "sing.npy"), lin._sing.get())
            np.save(os.path.join(args.save_factors, "X.npy"), X)
            np.save(os.path.join(args.save_factors, "Y.npy"), Y)
    model = (T.GaussBernoulliPrior(size=N, rho=args.rho, dtype=np_dt, ctx=ctx) @ T.SISOVariable(id="x")
             @ lin @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=Y.astype(np_dt), var=args.noise_var, dtype=np_dt, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    setup_s = time.perf_counter() - t_setup

    # ---------------- warm-up, then EXACTLY K timed steps
    never = 1 << 30
    ep.iterate(max_iter=max(args.warmup, 1), callback=T.PassCallback(), damping=args.damping, check_every=never)
    ctx._call("tramp_profile_enable", 1)
    barrier()
    t0 = time.perf_counter()
    ep.iterate(max_iter=args.steps, callback=T.PassCallback(), damping=args.damping, warm_start=True,
               check_every=never)
    barrier()
    elapsed = time.perf_counter() - t0
    import ctypes as C
    NT = len(TAGS)
    cnt = (C.c_int * NT)()
    ms = (C.c_double * NT)()
    by = (C.c_double * NT)()
    fl = (C.c_double * NT)()
    ctx._call("tramp_profile_report", cnt, ms, by, fl)
    ctx._call("tramp_profile_enable", 0)
    mse = float(np.mean(ep.mse("x", X)))
    if dist is not None:
        elapsed = ensemble.max_over_ranks(elapsed, device="cuda")

    if rank == 0:
        total_units = float(N) * B * world * args.steps
        kern = {TAGS[i]: {"launches": cnt[i], "ms": ms[i], "bytes": by[i], "flops": fl[i]} for i in range(NT)
                if cnt[i]}
        dom = max(kern, key=lambda k: kern[k]["ms"])
        d = kern[dom]
        # measured HBM traffic of the dominant kernel at this shape, if a PMC profile of it is committed.
        # both matrix passes of an iteration stream Vt (k x N): NT for t = Vt b, NN for the back-projection
        # (the Uk passes are gone: affine projection update + rotated output basis); the profile covers that shape.
        traffic = None
        traffic_detail = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01", "traffic.json")))["entries"]
            key = f"{dom}:{args.dtype}:R{M}:C{N}:B{B}"
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
                traffic_detail = {"algorithmic_bytes_per_launch": tj[key]["algorithmic_bytes"],
                                  "shape": f"{M}x{N} pass of the dominant kernel",
                                  "source": "profiles/r01/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                            "passes, FETCH_SIZE doubled per the gfx950 correction)"}
        except Exception:
            traffic = None
        use_mfma = B > 16 and dom.startswith("matvec")     # batch <= 16 streams at the HBM rate
        if use_mfma:
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS[args.dtype],
                    "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS[args.dtype], "traffic": traffic,
                    "traffic_detail": traffic_detail,
                    "peak_measured_issue_rate": MFMA_MEASURED_TFLOPS[args.dtype],
                    "peak_notes": "register-loop rates on this part (tools/probes): v_mfma_f64_16x16x4 47.5, "
                                  "v_mfma_f64_4x4x4_4b 76 TFLOP/s; PMC MfmaUtil of the f64 kernels 75.5 % "
                                  "(profiles/r01/pmc_mfma_util_kernels_N65536.csv)" if args.dtype == "f64" else None,
                    "avg_launch_ms": d["ms"] / d["launches"], "launches": d["launches"],
                    "hbm_GBps_algorithmic": d["bytes"] / (d["ms"] * 1e-3) / 1e9}
        else:
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_detail": traffic_detail,
                    "avg_launch_ms": d["ms"] / d["launches"],
                    "launches": d["launches"]}
        out = {
            "metric": "EP iterations/sec x N x instances (msg-components/sec), sparse GLM, alpha=0.5",
            "value": total_units / elapsed, "unit": "msg-components/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "sparse-GLM EP ensemble: GaussBernoulliPrior(rho=%g) + dense LinearChannel "
                                   "(Gaussian ensemble W shared by the instances) + GaussianLikelihood(var=%g), "
                                   "damping %g; dense W at N=1e6 (4 TB) cannot exist, this is BASELINE "
                                   "configs[3]'s per-GPU shard shape" % (args.rho, args.noise_var, args.damping),
                       "N": N, "M": M, "alpha": args.alpha, "instances_per_gpu": B,
                       "instances_total": B * world, "parallelism": f"ensemble-shard x{world}"},
            "roofline": roof,
            "kernels": kern,
            "setup": {"total_s": setup_s, "factorize_s": factorize_s,
                      "broadcast_W_s": bcast_s, "factors_loaded_from": args.load_factors},
            "mse_after_run": mse,
            "device": ctx.device_name(),
        }
        if world == 1 and not args.no_denoiser:
            out["denoiser_n1e6"] = denoiser_roofline(T, ctx, args.dtype)
            try:
                out["structured_n1e6"] = structured_n1e6(T, ctx, args.dtype)
            except Exception as e:                                            # noqa: BLE001
                out["structured_n1e6"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    """Single-rank runs retry at half the size if the setup fails (e.g. the eigensolver or an
    allocation): a smaller honest number beats no number; the size actually run is in `config`."""
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", 1))
    while True:
        try:
            return run(args)
        except Exception as e:      # noqa: BLE001
            if world > 1 or args.n <= 8192 or args.load_factors:
                raise
            import gc
            import traceback
            traceback.print_exc(file=sys.stderr)
            print(f"bench.py: run at N={args.n} failed ({e!r}); retrying at N={args.n // 2}", file=sys.stderr)
            gc.collect()
            args.n //= 2


if __name__ == "__main__":
    main()
