"""Ensemble axis across GPUs: partition independent problem instances over ranks and share
the sensing matrix with ONE broadcast at setup.

Instances never interact, so there is no per-iteration collective and no cross-GPU
reduction; `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU
tests) is plumbing for the setup broadcast, the barriers and the final gather only.
The reference runs the same ensemble as a sequential Python loop
(tramp/experiments/multiple_experiments.py:30-49).
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous block [start, stop) of `total` instances owned by `rank` (sizes differ by <= 1)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def instance_seeds(base_seed, start, stop):
    "Per-instance RNG seeds: an instance draws the same data whatever the number of ranks."
    return [int(base_seed) + 1000003 * i for i in range(start, stop)]


def broadcast_tensor(tensor, src=0, group=None):
    """In-place broadcast of the shared matrix from `src`; returns seconds spent.
    `tensor` is a torch tensor on this rank's device (CPU tensor under gloo)."""
    import time
    import torch
    import torch.distributed as dist
    if tensor.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(tensor, src=src, group=group)
    if tensor.is_cuda:
        torch.cuda.synchronize()
    return time.perf_counter() - t0


def gather_to_root(values, dst=0, group=None):
    """Gather per-rank result arrays (n_iter, v, mse per instance) on `dst`; tiny, end of run."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = [None] * world if dist.get_rank(group) == dst else None
    dist.gather_object(np.asarray(values), out, dst=dst, group=group)
    return out


def max_over_ranks(x, device=None, group=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
