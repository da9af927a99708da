#!/usr/bin/env python3
"""Golden vectors for the complex variants (``real=False``) of the Fourier-diagonal channels:
ConvChannel (tramp/channels/linear/conv_channel.py:25-32, 81-109) and DFTChannel (dft_channel.py:31-67).
Vectors are the reference's planar arrays (2,) + shape.

Runs ONLY in the build container (imports /root/reference); writes tests/golden/conv_complex.npz - inputs and the
reference's outputs, nothing else.

    python tools/make_golden_conv_complex.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, os.environ.get("TRAMP_REFERENCE", "/root/reference"))

from tramp.channels import ConvChannel, DFTChannel  # noqa: E402


def main():
    rs = np.random.RandomState(20260304)
    conv, dft = [], []
    for shape in [(16,), (15,), (64,), (6, 8), (8, 16), (16, 32), (4, 4, 16), (3, 4, 5)]:
        f = np.zeros(shape, dtype=complex)
        idx = tuple(slice(0, min(3, s)) for s in shape)
        f[idx] = rs.standard_normal(f[idx].shape) + 1j * rs.standard_normal(f[idx].shape)
        ch = ConvChannel(f, real=False)
        for az, ax in [(1.3, 0.7), (0.2, 5.0)]:
            bz, bx = rs.standard_normal((2,) + shape), rs.standard_normal((2,) + shape)
            rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
            rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
            a_b, b_b = ch.compute_backward_message(az, bz, ax, bx)
            a_f, b_f = ch.compute_forward_message(az, bz, ax, bx)
            conv.append(dict(shape=shape, filter=f, az=az, ax=ax, bz=bz, bx=bx, rz=rz, vz=float(vz), rx=rx, vx=float(vx),
                             a_bwd=float(a_b), b_bwd=b_b, a_fwd=float(a_f), b_fwd=b_f, sample=ch.sample(bz),
                             n_eff=float(ch.compute_n_eff(az, ax)), tau=float(ch.second_moment(1.7))))
        d = DFTChannel(real=False)
        bz, bx = rs.standard_normal((2,) + shape), rs.standard_normal((2,) + shape)
        a_f, b_f = d.compute_forward_message(1.3, bz, 0.7, bx)
        a_b, b_b = d.compute_backward_message(1.3, bz, 0.7, bx)
        dft.append(dict(shape=shape, bz=bz, bx=bx, a_fwd=float(a_f), b_fwd=b_f, a_bwd=float(a_b), b_bwd=b_b,
                        sample=d.sample(bz)))
    np.savez_compressed(os.path.join(OUT, "conv_complex.npz"), data=np.array(conv, dtype=object),
                        dft=np.array(dft, dtype=object))
    print(len(conv), "conv cases,", len(dft), "dft cases")


if __name__ == "__main__":
    main()
