"""Worker of tests/test_ensemble_dist.py: world_size-2 gloo run of the ensemble helpers."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramp_amd import ensemble  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 7
    start, stop = ensemble.shard_range(total, rank, world)
    # rank 0 draws W, everyone else receives it in one broadcast
    M, N = 6, 10
    W = torch.zeros((M, N), dtype=torch.float64)
    if rank == 0:
        W.copy_(torch.from_numpy(np.random.RandomState(5).randn(M, N)))
    secs = ensemble.broadcast_tensor(W, src=0)
    assert secs >= 0
    expect = np.random.RandomState(5).randn(M, N)
    assert np.array_equal(W.numpy(), expect), "broadcast W differs"
    # per-instance work: seeds do not depend on the partition
    seeds = ensemble.instance_seeds(42, start, stop)
    local = np.array([np.random.RandomState(s).randn() for s in seeds])
    gathered = ensemble.gather_to_root(local)
    t = ensemble.max_over_ranks(1.0 + rank)
    assert t == float(world)
    if rank == 0:
        allv = np.concatenate(gathered)
        ref = np.array([np.random.RandomState(s).randn() for s in ensemble.instance_seeds(42, 0, total)])
        assert np.array_equal(allv, ref), (allv, ref)
        print("DIST_OK", total, world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
