#!/usr/bin/env python3
"""Golden vectors for the log-partitions of the rest of the piecewise-linear family (SURVEY section 8f rank 2 + 5):
LeakyRelu / AsymmetricAbs / HardTanh / HardSigmoid / SymmetricDoor channels
(tramp/channels/activation/piecewise_linear_channel.py:55-58, 72-132) and the Relu / LeakyRelu / HardTanh /
HardSigmoid / SymmetricDoor / AsymmetricAbs likelihoods (tramp/likelihoods/piecewise_linear_likelihood.py:216-224,
260-320), computed by importing the reference.

Runs ONLY in the build container (imports /root/reference); writes tests/golden/f2_logz.npz: inputs and the
reference's outputs, nothing else.  For AsymmetricAbsLikelihood the reference's outputs are NaN for every input
(np.log of the descending branch's slope -1, :59): they are recorded as found.

    python tools/make_golden_f2.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, os.environ.get("TRAMP_REFERENCE", "/root/reference"))

from tramp.channels import (LeakyReluChannel, AsymmetricAbsChannel, HardTanhChannel, HardSigmoidChannel,  # noqa: E402
                            SymmetricDoorChannel)
from tramp.likelihoods.piecewise_linear_likelihood import (ReluLikelihood, LeakyReluLikelihood,  # noqa: E402
                                                           HardTanhLikelihood, HardSigmoidLikelihood,
                                                           SymmetricDoorLikelihood, AsymmetricAbsLikelihood)


def main():
    rs = np.random.RandomState(20261003)
    n = 301
    cases = []
    chans = [("l-relu", dict(slope=0.1), LeakyReluChannel), ("a-abs", dict(shift=1e-4), AsymmetricAbsChannel),
             ("a-abs", dict(shift=0.3), AsymmetricAbsChannel), ("a-abs", dict(shift=-0.2), AsymmetricAbsChannel),
             ("h-tanh", {}, HardTanhChannel), ("h-sigm", {}, HardSigmoidChannel),
             ("door", dict(width=0.7), SymmetricDoorChannel)]
    for name, kw, cls in chans:
        ch = cls(**kw)
        for az, ax, sb in [(1.3, 0.7, 1.5), (0.4, 2.5, 3.0), (3.0, 0.05, 0.5)]:
            bz = sb * rs.standard_normal(n)
            bx = sb * rs.standard_normal(n)
            with np.errstate(all="ignore"):
                A = ch.compute_log_partition(az, bz, ax, bx)
            cases.append(dict(family="channel", name=name, kwargs=kw, az=az, ax=ax, bz=bz, bx=bx, A=float(A)))
    liks = [("relu", {}, ReluLikelihood), ("l-relu", dict(slope=0.1), LeakyReluLikelihood),
            ("h-tanh", {}, HardTanhLikelihood), ("h-sigm", {}, HardSigmoidLikelihood),
            ("door", dict(width=0.7), SymmetricDoorLikelihood), ("a-abs", dict(shift=1e-4), AsymmetricAbsLikelihood),
            ("a-abs", dict(shift=0.3), AsymmetricAbsLikelihood)]
    for name, kw, cls in liks:
        for az, sb in [(1.3, 1.5), (0.4, 3.0)]:
            z = 1.5 * rs.standard_normal(n)
            y = cls(y=z, **kw).sample(z)           # an output the factor can produce
            if name == "a-abs":
                y = np.abs(z)
                y[:5] = [0.0, -0.4, 1e-5, kw["shift"], -kw["shift"]]       # compatible with one branch, with none
            bz = sb * rs.standard_normal(n)
            lk = cls(y=y, **kw)
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                A = lk.compute_log_partition(az, bz, y)
                r, v = lk.compute_backward_posterior(az, bz, y)
            cases.append(dict(family="likelihood", name=name, kwargs=kw, az=az, bz=bz, y=y, A=float(A),
                              r=np.asarray(r), v=float(v)))
    np.savez(os.path.join(OUT, "f2_logz.npz"), cases=np.array(cases, dtype=object))
    for c in cases:
        print(c["family"], c["name"], c["kwargs"], "A =", c["A"])


if __name__ == "__main__":
    main()
