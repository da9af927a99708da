#!/usr/bin/env python3
"""Golden vectors for LinearChannel(W, precompute_svd=False) (tramp/channels/linear/linear_channel.py:43-46, 79-82):
tall, wide (Nz > rank: `singular` is cut from the ascending spectrum there) and rank-deficient matrices.

Runs ONLY in the build container (imports /root/reference); writes tests/golden/solve.npz - inputs and the
reference's outputs (spectrum, rank, n_eff, posteriors, messages, log-partition), nothing else.

    python tools/make_golden_solve.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, os.environ.get("TRAMP_REFERENCE", "/root/reference"))

from tramp.channels import LinearChannel  # noqa: E402


def main():
    rs = np.random.RandomState(20260303)
    cases = []
    mats = {"tall_12x8": rs.randn(12, 8) / np.sqrt(8), "wide_8x12": rs.randn(8, 12) / np.sqrt(12),
            "square_10": rs.randn(10, 10) / np.sqrt(10), "wide_30x64": rs.randn(30, 64) / 8.0}
    low = rs.randn(9, 3) @ rs.randn(3, 7)                      # rank 3, 9 x 7
    mats["rank3_9x7"] = low
    for name, W in mats.items():
        ch = LinearChannel(W, precompute_svd=False)
        Nx, Nz = W.shape
        for az, ax in [(1.3, 0.7), (0.2, 5.0), (3.0, 1e-3), (1e-6, 2.0)]:
            bz, bx = rs.randn(Nz), rs.randn(Nx)
            rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
            rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
            a_b, b_b = ch.compute_backward_message(az, bz, ax, bx)
            a_f, b_f = ch.compute_forward_message(az, bz, ax, bx)
            cases.append(dict(name=name, W=W, az=az, ax=ax, bz=bz, bx=bx, rank=int(ch.rank), spectrum=ch.spectrum.copy(),
                              n_eff=float(ch.compute_n_eff(az, ax)), rz=rz, vz=float(vz), rx=rx, vx=float(vx),
                              a_bwd=float(a_b), b_bwd=b_b, a_fwd=float(a_f), b_fwd=b_f,
                              logZ=float(ch.compute_log_partition(az, bz, ax, bx))))
    np.savez_compressed(os.path.join(OUT, "solve.npz"), data=np.array(cases, dtype=object))
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
