"""Multi-GPU path on CPU: partition logic, and a world_size-2 gloo run of the setup
broadcast / gather helpers that bench.py uses with RCCL."""
import os
import subprocess
import sys

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


def test_world_size_2_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "DIST_OK 7 2" in p.stdout
