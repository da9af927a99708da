"""Multi-GPU path on CPU: partition logic, the rendezvous, and a world_size-2 run of the ensemble
path (broadcast of W, per-instance EP, gather) on the socket stub of the communicator API, compared
instance by instance with the 1-rank run."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from tramp_amd import ensemble

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("total,world", [(512, 8), (7, 2), (3, 8), (0, 4), (64, 1)])
def test_shard_range_partitions(total, world):
    ranges = [ensemble.shard_range(total, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == total
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in ranges]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total
    with pytest.raises(ValueError):
        ensemble.shard_range(total, world, world)


def test_instance_seeds_independent_of_partition():
    whole = ensemble.instance_seeds(9, 0, 10)
    parts = sum((ensemble.instance_seeds(9, *ensemble.shard_range(10, r, 3)) for r in range(3)), [])
    assert whole == parts and len(set(whole)) == 10


def test_rendezvous_exchange_threads():
    "3 ranks in one process: all-gather of byte strings, broadcast, gather, max on the socket stub"
    port = ensemble.free_port()
    world, out, errs = 3, {}, []

    def run(rank):
        try:
            comm = ensemble.SocketComm(ensemble.Rendezvous(rank, world, "127.0.0.1", port, timeout=60))
            a = np.arange(6, dtype=np.float64).reshape(2, 3) * (rank == 1)
            comm.broadcast(a, root=1)
            g = comm.gather_scalars([rank, 10.0 * rank])
            out[rank] = (a.copy(), g, comm.max(rank * 2.5), comm.rdv.exchange(bytes([rank]) * (rank + 1)))
            comm.barrier()
            comm.close()
        except Exception as e:       # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(90) for t in ts]
    assert not errs, errs
    for r in range(world):
        a, g, m, parts = out[r]
        assert np.array_equal(a, np.arange(6.0).reshape(2, 3))
        assert np.array_equal(g, np.array([[0, 0], [1, 10], [2, 20]], dtype=np.float64))
        assert m == 5.0
        assert parts == [b"\x00", b"\x01\x01", b"\x02\x02\x02"]


class _FakeRcclCtx:
    """A device context whose RCCL entry points behave as told, on this rank only:
    "no_lib"  the probe (tramp_comm_available) fails - library missing;
    "no_init" the probe passes, tramp_comm_init fails (IPC refused, duplicate device);
    "ok"      tramp_comm_init hands out a communicator handle.
    Records every call; tramp_comm_destroy is counted wherever it comes from (the _call path or __del__'s
    direct library call)."""
    def __init__(self, mode):
        self.mode = mode
        self.calls = []
        self.handle = 1
        self.lib = self

    def tramp_comm_destroy(self, ctx_handle, h):         # RcclComm.__del__ goes through ctx.lib
        self.calls.append("tramp_comm_destroy")
        return 0

    def _call(self, name, *args):
        self.calls.append(name)
        if name == "tramp_comm_available":
            if self.mode == "no_lib":
                raise RuntimeError(name + ": librccl not available")
            return 0
        if name == "tramp_comm_unique_id":
            return 0
        if name == "tramp_comm_init":
            if self.mode == "ok":
                args[3]._obj.value = 0xC0FFEE             # C.byref(handle)
                return 0
            raise RuntimeError(name + ": hipIpcGetMemHandle: invalid argument")
        if name == "tramp_comm_destroy":
            return 0
        raise RuntimeError(name + ": no GPU in this test")


@pytest.mark.parametrize("modes", [("no_lib", "no_lib", "no_lib"), ("no_lib", "ok", "ok"), ("ok", "ok", "no_lib"),
                                   ("no_init", "no_init", "no_init"), ("ok", "no_init", "ok"), ("no_init", "ok", "ok")])
def test_rccl_failure_falls_back_to_sockets_on_every_rank(modes):
    """whichever rank cannot build the RCCL communicator, ALL ranks end up on the socket communicator, in step
    (the bench then computes the shared factors on every rank instead of receiving a broadcast).  A missing
    library is found by the probe, before any rank enters the collective init; a communicator that did come up
    on a healthy rank is destroyed exactly once."""
    import gc
    port = ensemble.free_port()
    world, out, errs, logs = 3, {}, [], []
    ctxs = [_FakeRcclCtx(m) for m in modes]

    def run(rank):
        try:
            # env_world reads os.environ, shared by the threads: hand each thread its own view
            comm = _init_with_env(lambda: (rank, rank, world, "127.0.0.1", port), ctxs[rank], logs)
            out[rank] = (comm.backend, comm.max(float(rank)), comm.gather_scalars([rank]).ravel().tolist())
            comm.barrier()
            comm.close()
        except Exception as e:       # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(90) for t in ts]
    assert not errs, errs
    assert all(out[r] == ("socket", 2.0, [0.0, 1.0, 2.0]) for r in range(world)), out
    assert logs and "socket communicator" in logs[0]
    gc.collect()
    for c in ctxs:
        if "no_lib" in modes:
            assert "tramp_comm_init" not in c.calls and "tramp_comm_unique_id" not in c.calls, c.calls
        inits_ok = c.mode == "ok" and "tramp_comm_init" in c.calls
        assert c.calls.count("tramp_comm_destroy") == (1 if inits_ok else 0), (c.mode, c.calls)


def _init_with_env(env_fn, ctx, logs):
    "init_comm_or_sockets with a per-thread env_world"
    import types
    mod = types.ModuleType("ensemble_view")
    mod.__dict__.update(ensemble.__dict__)
    mod.env_world = env_fn
    fn = types.FunctionType(ensemble.init_comm_or_sockets.__code__, mod.__dict__)
    return fn(ctx, timeout=60.0, log=logs.append)


def _run_world(world):
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    procs = []
    port = ensemble.free_port()
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py")], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-2000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("DIST_RESULT ")]
    assert line, outs[0][0][-2000:]
    return np.array(json.loads(line[0][len("DIST_RESULT "):]))


def test_world_size_2_matches_single_rank():
    """every instance's (n_iter, v, mse) from the 2-rank run equals the 1-rank run's, bit for bit"""
    one = _run_world(1)
    two = _run_world(2)
    assert one.shape == two.shape == (5, 3)
    assert np.array_equal(one, two), (one, two)
    assert np.all(one[:, 0] > 3) and np.all(one[:, 2] < 0.5)


def test_launcher_returns_when_a_rank_dies(tmp_path):
    "one rank exits with an error while the other would wait forever: the launcher stops it and reports failure"
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                      "time.sleep(600)\n")
    t0 = time.time()
    rc = ensemble.launch_ranks(2, [str(script)], grace=1.0)
    assert rc != 0 and time.time() - t0 < 60


def test_bench_rank_logic_world_8():
    """bench.py's rank logic at the world size of BASELINE configs[3] (8 ranks x 64 instances = 512) on the socket
    communicator: every instance draws the same signal whatever the partition (the reference's sequential loop over
    seeds, tramp/experiments/multiple_experiments.py:30-49), and rank 0's gathered per-instance results come back in
    instance order."""
    import argparse
    import bench
    world, B, N, M = 8, 64, 24, 12
    args = argparse.Namespace(rho=0.3)
    X1, n1 = bench.teacher_signals(ensemble, args, 0, world * B, N, M, np.float64)
    assert len({tuple(r) for r in X1.round(12).tolist()}) == world * B            # 512 different instances
    port = ensemble.free_port()
    out, errs = {}, []

    def run(rank):
        try:
            comm = ensemble.SocketComm(ensemble.Rendezvous(rank, world, "127.0.0.1", port, timeout=60))
            start, stop = ensemble.shard_range(world * B, rank, world)
            assert stop - start == B
            X, noise = bench.teacher_signals(ensemble, args, start, stop, N, M, np.float64)
            a, b = bench.gather_instances(comm, [X.sum(axis=1), noise[:, 0]], world, B)
            out[rank] = (X, noise, a, b, comm.max(float(rank)))
            comm.barrier()
            comm.close()
        except Exception as e:       # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errs, errs
    for r in range(world):
        X, noise, a, b, mx = out[r]
        assert np.array_equal(X, X1[r * B:(r + 1) * B]) and np.array_equal(noise, n1[r * B:(r + 1) * B])
        assert np.array_equal(a, X1.sum(axis=1)) and np.array_equal(b, n1[:, 0]) and mx == world - 1.0
