#!/usr/bin/env python3
"""Golden vectors for the diagonal-belief (``isotropic=False``) variants of the separable factors: the mode every
factor-level test of the reference runs in (tramp/tests/test_priors.py:17-51, tramp/tests/test_likelihoods.py:22-83:
size 100, a = linspace(1, 2, 100), b = linspace(-2, 2, 100), y = f(linspace(-3, 3, 100))).

Runs ONLY in the build container (imports /root/reference); writes tests/golden/diag.npz - inputs and the
reference's outputs (posterior r, v arrays; message a_new, b_new arrays with the element-wise clip of
tramp/base.py:250-255; mean log-partition A), nothing else.

    python tools/make_golden_diag.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, os.environ.get("TRAMP_REFERENCE", "/root/reference"))

from tramp.priors import GaussianPrior, GaussBernoulliPrior, BinaryPrior, MAP_L1NormPrior  # noqa: E402
from tramp.likelihoods import (GaussianLikelihood, AbsLikelihood, SgnLikelihood, ReluLikelihood,  # noqa: E402
                               LeakyReluLikelihood, HardTanhLikelihood, HardSigmoidLikelihood,
                               SymmetricDoorLikelihood)
from tramp.utils.misc import relu, leaky_relu, hard_tanh, hard_sigm, symm_door  # noqa: E402


def prior_case(cls, name, kwargs, a, b):
    p = cls(size=b.shape[0], isotropic=False, **kwargs)
    with np.errstate(all="ignore"):
        r, v = p.compute_forward_posterior(a, b)
        an, bn = p.compute_forward_message(a, b)
        A = p.compute_log_partition(a, b)
    assert np.shape(v) == b.shape
    return dict(family="prior", name=name, kwargs=kwargs, a=a, b=b, r=r, v=v, a_new=np.asarray(an), b_new=np.asarray(bn),
                A=float(A))


def lik_case(cls, name, kwargs, y, a, b):
    lk = cls(y=y, isotropic=False, **kwargs)
    with np.errstate(all="ignore"):
        r, v = lk.compute_backward_posterior(a, b, y)
        an, bn = lk.compute_backward_message(a, b)
        A = lk.compute_log_partition(a, b, y)
    assert np.shape(v) == b.shape
    return dict(family="likelihood", name=name, kwargs=kwargs, y=y, a=a, b=b, r=r, v=v, a_new=np.asarray(an),
                b_new=np.asarray(bn), A=float(A))


def main():
    rs = np.random.RandomState(20260301)
    cases = []
    # the reference's own inputs
    a100, b100 = np.linspace(1, 2, 100), np.linspace(-2, 2, 100)
    z = np.linspace(-3, 3, 100)
    # ragged random inputs; and precisions that hit both ends of the clip [1e-11, 1e11]
    n = 257
    ar, br = rs.uniform(0.3, 4.0, n), 2.5 * rs.standard_normal(n)
    zr = 1.5 * rs.standard_normal(n)
    ae = np.array([1e-12, 1e-11, 1e-6, 1e-3, 1.0, 1e3, 1e6, 1e11, 1e12, 2.0])
    be = np.array([0.0, 1e-5, -1e-3, 0.5, -2.0, 40.0, -1e3, 1e9, 1e11, 0.0])
    for tag, a, b in (("ref", a100, b100), ("rand", ar, br), ("edge", ae, be)):
        for cls, name, kw in ((GaussianPrior, "gaussian", {}), (GaussianPrior, "gaussian", dict(mean=0.3, var=2.0)),
                              (GaussBernoulliPrior, "gauss_bernoulli", {}),
                              (GaussBernoulliPrior, "gauss_bernoulli", dict(rho=0.1, mean=0.2, var=1.5)),
                              (BinaryPrior, "binary", {}), (BinaryPrior, "binary", dict(p_pos=0.3)),
                              (MAP_L1NormPrior, "l1", {}), (MAP_L1NormPrior, "l1", dict(gamma=0.5))):
            c = prior_case(cls, name, kw, a, b)
            c["inputs"] = tag
            cases.append(c)
    for tag, a, b, zz in (("ref", a100, b100, z), ("rand", ar, br, zr)):
        for cls, name, kw, y in (
                (GaussianLikelihood, "gaussian", {}, zz), (GaussianLikelihood, "gaussian", dict(var=0.3), zz),
                (AbsLikelihood, "abs", {}, np.abs(zz)), (SgnLikelihood, "sgn", {}, np.sign(zz)),
                (ReluLikelihood, "relu", {}, relu(zz)),
                (LeakyReluLikelihood, "l-relu", dict(slope=0.1), leaky_relu(zz, 0.1)),
                (HardTanhLikelihood, "h-tanh", {}, hard_tanh(zz)),
                (HardSigmoidLikelihood, "h-sigm", {}, hard_sigm(zz)),
                (SymmetricDoorLikelihood, "door", dict(width=0.1), symm_door(zz, 1.)),     # as the reference's test
                (SymmetricDoorLikelihood, "door", dict(width=1.0), symm_door(zz, 1.))):
            c = lik_case(cls, name, kw, y, a, b)
            c["inputs"] = tag
            cases.append(c)
    path = os.path.join(OUT, "diag.npz")
    np.savez_compressed(path, data=np.array(cases, dtype=object))
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path) / 1024:.1f} KiB")
    for c in cases:
        bad = sum(int(np.sum(~np.isfinite(np.asarray(c[k], dtype=float)))) for k in ("r", "v", "a_new", "b_new"))
        print(f"  {c['family']:10s} {c['name']:16s} {c['inputs']:5s} {c['kwargs']}  non-finite outputs: {bad}")


if __name__ == "__main__":
    main()
