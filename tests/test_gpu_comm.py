"""The RCCL leg of the C ABI on one GPU (world of 1: every collective is the identity, but the whole
chain dlopen(librccl) -> ncclGetUniqueId -> ncclCommInitRank -> ncclBroadcast / ncclAllGather /
ncclAllReduce runs on the context's stream).  The multi-rank logic is covered on CPU by
tests/test_ensemble_dist.py with the socket stub of the same Python API."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_world_of_one(ctx):
    from tramp_amd import ensemble
    comm = ensemble.RcclComm(ctx, ensemble.Rendezvous(0, 1))
    assert (comm.rank, comm.world) == (0, 1)
    a = ctx.randn((3, 1000), np.float64, seed=1)
    ref = a.get()
    secs = comm.broadcast(a, root=0)
    assert secs >= 0 and np.array_equal(a.get(), ref)
    g = comm.gather_scalars([1.5, -2.0, 7.0])
    assert g.shape == (1, 3) and np.array_equal(g[0], [1.5, -2.0, 7.0])
    assert comm.max(3.25) == 3.25
    comm.barrier()
    comm.close()
