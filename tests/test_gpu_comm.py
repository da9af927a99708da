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


def test_broadcast_of_a_factor_sized_buffer_in_pieces(ctx):
    """The shared matrix's thin factors are 20 - 60 GB per piece (BASELINE configs[3]): tramp_bcast_matrix walks such a
    buffer in <= 1 GiB collectives, so byte offsets pass 2^31 and 2^32 and the last piece is ragged.  In a world of
    one the broadcast must leave every byte where it was: 8 GiB + 12345 bytes, sentinels at both sides of every
    piece boundary and of the 2^31 / 2^32 offsets, checked afterwards."""
    from tramp_amd import ensemble
    free, _total = ctx.mem_info()
    nbytes = (8 << 30) + 12345
    if free < nbytes + (1 << 30):
        pytest.skip("not enough free HBM for an 8 GiB buffer")
    comm = ensemble.RcclComm(ctx, ensemble.Rendezvous(0, 1))
    buf = ctx.empty((nbytes,), np.uint8)
    ctx._call("tramp_fill", 1, buf.ptr, nbytes // 8, 3.0)                     # f64 3.0 pattern over the whole buffer
    marks = sorted({o + d for o in [0, 1 << 30, 1 << 31, (1 << 32), 5 << 30, 8 << 30, nbytes - 64]
                    for d in (-64, 0) if 0 <= o + d <= nbytes - 64})
    rs = np.random.RandomState(0)
    pats = {o: rs.randint(0, 256, 64).astype(np.uint8) for o in marks}
    for o, p in pats.items():
        ctx._call("tramp_h2d", buf.ptr + o, p.ctypes.data, 64)
    secs = comm.broadcast(buf, root=0)
    assert secs > 0
    got = np.empty(64, np.uint8)
    for o, p in pats.items():
        ctx._call("tramp_d2h", got.ctypes.data, buf.ptr + o, 64)
        assert np.array_equal(got, p), f"bytes at offset {o} changed"
    mid = np.empty(8, np.float64)
    for o in ((3 << 30) + 8 * 1000, (7 << 30) + 8 * 77):
        ctx._call("tramp_d2h", mid.ctypes.data, buf.ptr + o, 64)
        assert (mid == 3.0).all()
    comm.close()
