"""Worker of tests/test_ensemble_dist.py: one rank of a world_size-2 CPU run of the ensemble path.

Same partition / broadcast / gather calls as bench.py, on the socket stub of the communicator API
(`ensemble.SocketComm`; on the GPU box the same calls go to RCCL through the C ABI).  The per-instance
solver is the oracle's EP (test infrastructure): what is checked is that an instance gives the same
(n_iter, v, mse) whatever rank it lands on."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tramp_amd import ensemble  # noqa: E402
from oracle import tramp_oracle as orc  # noqa: E402

TOTAL, N, M, RHO, NV = 5, 48, 24, 0.2, 1e-2


def solve_instances(W, start, stop):
    "per-instance EP on instances [start, stop) of the ensemble: rows of (n_iter, v, mse)"
    lin = orc.LinearFactor(W)
    out = []
    for seed in ensemble.instance_seeds(77, start, stop):
        rs = np.random.RandomState(seed % (2**31))
        x = orc.gb_sample(rs, N, RHO)
        y = W @ x + np.sqrt(NV) * rs.standard_normal(M)
        res = orc.ep_chain([orc.GBPriorFactor(N, rho=RHO), lin, orc.GaussLikFactor(y, NV)], [N, M],
                           max_iter=200, damping=0.1, tol=1e-6)
        out.append([float(res["n_iter"]), float(res["v"][0]), float(np.mean((res["r"][0] - x)**2))])
    return np.array(out).reshape(-1, 3)


def main():
    comm = ensemble.init_comm(backend="socket", timeout=120)
    rank, world = comm.rank, comm.world
    start, stop = ensemble.shard_range(TOTAL, rank, world)
    # rank 0 draws W, everyone else receives it in one broadcast
    W = np.zeros((M, N))
    if rank == 0:
        W[...] = orc.gaussian_ensemble(np.random.RandomState(5), M, N)
    secs = comm.broadcast(W, root=0)
    assert secs >= 0
    assert np.array_equal(W, orc.gaussian_ensemble(np.random.RandomState(5), M, N)), "broadcast W differs"
    comm.barrier()
    local = solve_instances(W, start, stop)
    # ranks own different numbers of instances: pad to the largest shard, as bench.py does
    width = -(-TOTAL // world)
    padded = np.full((width, 3), np.nan)
    padded[:len(local)] = local
    gathered = comm.gather_scalars(padded.ravel()).reshape(world, width, 3)
    t = comm.max(1.0 + rank)
    assert t == float(world)
    if rank == 0:
        rows = np.concatenate([gathered[r][:ensemble.shard_range(TOTAL, r, world)[1] - ensemble.shard_range(TOTAL, r, world)[0]]
                               for r in range(world)])
        print("DIST_RESULT " + json.dumps(rows.tolist()))
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
