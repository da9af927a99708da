#!/usr/bin/env python3
"""Generate golden vectors by importing the reference (sphinxteam/tramp).

Runs ONLY in the build container (needs /root/reference); its outputs, small
``.npz`` files of inputs and expected outputs, are committed under
``tests/golden/``.  Nothing here travels to the GPU box except those data files.

The reference's graph layer needs networkx 1.x; a test-time shim
below restores the list-returning API before ``tramp`` is imported.

    python tools/make_golden.py [--skip-slow]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
REF = os.environ.get("TRAMP_REFERENCE", "/root/reference")


def install_networkx1_shim():
    import networkx as nx

    class DiGraph1(nx.DiGraph):
        @property
        def node(self):
            return self._node

        def predecessors(self, n):
            return list(super().predecessors(n))

        def successors(self, n):
            return list(super().successors(n))

        def in_edges(self, *a, **k):
            return list(super().in_edges(*a, **k))

        def edges(self, *a, **k):
            return list(super().edges(*a, **k))

        def nodes(self, *a, **k):
            return list(super().nodes(*a, **k))

    _topo = nx.topological_sort
    nx.DiGraph = DiGraph1
    nx.topological_sort = lambda g: list(_topo(g))


install_networkx1_shim()
sys.path.insert(0, REF)

from tramp.priors import GaussBernoulliPrior, MAP_L1NormPrior, GaussianPrior  # noqa: E402
from tramp.likelihoods import GaussianLikelihood, SgnLikelihood, AbsLikelihood  # noqa: E402
from tramp.likelihoods import ReluLikelihood, LeakyReluLikelihood  # noqa: E402
from tramp.likelihoods import HardTanhLikelihood, HardSigmoidLikelihood, SymmetricDoorLikelihood  # noqa: E402
from tramp.channels import (  # noqa: E402
    LinearChannel, SgnChannel, AbsChannel, ReluChannel, MarchenkoPasturChannel,
    LeakyReluChannel, AsymmetricAbsChannel, HardTanhChannel, HardSigmoidChannel, SymmetricDoorChannel
)
from tramp.channels import (  # noqa: E402
    ConvChannel, Blur1DChannel, Blur2DChannel, DifferentialChannel, LaplacianChannel
)
from tramp.channels import DFTChannel, RotationChannel  # noqa: E402
from tramp.variables import SISOVariable as V  # noqa: E402
from tramp.algos import ExpectationPropagation, StateEvolution, EarlyStoppingEP  # noqa: E402
from tramp.algos import EarlyStopping, ConstantInit  # noqa: E402
from tramp.base import Variable  # noqa: E402
from tramp.utils.truncated_normal import log_Phi  # noqa: E402
from tramp.priors.map_L1_norm_prior import v_soft_threshold  # noqa: E402


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


# ---------------------------------------------------------------- factors
def message_law(rs, a, x):
    "Bayes-optimal message b = a x + sqrt(a) xi (checks/check_high_dim.py:11-13)."
    return a * x + np.sqrt(a) * rs.standard_normal(x.shape)


def factor_goldens():
    rs = np.random.RandomState(20260101)
    N = 257                       # ragged: not a multiple of 64 / 256
    out = {}

    # SURVEY section 8c known-answer inputs
    kat_b = np.array([-3, -.5, 0, .25, 2, 6.])
    kat_bx = np.array([.5, -1, 2, 0, -.3, 1.5])
    kat_y = np.array([1, -1, 1, 1, -1, 1.])
    out["kat_b"], out["kat_bx"], out["kat_y"] = kat_b, kat_bx, kat_y

    # --- Gauss-Bernoulli prior
    gb_cases = []
    x = rs.standard_normal(N) * rs.binomial(1, 0.1, N)
    b_rand = message_law(rs, 1.3, x)
    b_edge = np.array([0., 1e-300, -1e-300, 50., -50., 700., -700., 1e-8, 3.5, -3.5])
    for (rho, mean, var, ax, b) in [
        (0.1, 0., 1., 1.3, kat_b),
        (0.1, 0., 1., 1.3, b_rand),
        (0.05, 0., 1., 0., b_rand),           # first iteration: ax = 0
        (0.05, 0., 1., 0, np.zeros(16)),      # python int 0, b = 0
        (0.5, 0.3, 2.0, 0.7, b_rand),
        (0.01, -1.0, 0.25, 25.0, b_rand * 10),
        (0.1, 0., 1., 1.3, b_edge),
        (0.3, 0.1, 1.5, 1e-11, b_rand),
        (0.3, 0.1, 1.5, 1e+11, b_rand * 1e11),
    ]:
        p = GaussBernoulliPrior(size=b.shape[0], rho=rho, mean=mean, var=var)
        r, v = p.compute_forward_posterior(ax, b)
        an, bn = p.compute_forward_message(ax, b)
        pd_ = GaussBernoulliPrior(size=b.shape[0], rho=rho, mean=mean, var=var,
                                  isotropic=False)
        rd, vd = pd_.compute_forward_posterior(ax, b)
        gb_cases.append(dict(rho=rho, mean=mean, var=var, ax=float(ax), b=b, r=r,
                             v=v, a_new=an, b_new=bn, v_diag=vd, eta=p.eta))
    # diagonal beliefs with array ax (reference tests/test_priors.py:17-24)
    axd = np.linspace(1, 2, 100)
    bxd = np.linspace(-2, 2, 100)
    pd_ = GaussBernoulliPrior(size=100, rho=0.5, isotropic=False)
    rd, vd = pd_.compute_forward_posterior(axd, bxd)
    and_, bnd = pd_.compute_forward_message(axd, bxd)
    out["gb_diag"] = dict(ax=axd, bx=bxd, r=rd, v=vd, a_new=and_, b_new=bnd)
    out["gb"] = gb_cases

    # --- MAP L1 prior
    l1_cases = []
    gam = 1.0
    b_l1_edge = np.array([gam, -gam, np.nextafter(gam, 2), np.nextafter(gam, 0),
                          -np.nextafter(gam, 2), 0., -0., 1e-310, 5., -5., 1e300])
    for (gamma, ax, b) in [(1.0, 1.3, kat_b), (1.0, 1.3, b_rand), (0.37, 2.5, b_rand),
                           (1.0, 0.9, b_l1_edge), (2.0, 1e-3, b_rand)]:
        p = MAP_L1NormPrior(size=b.shape[0], gamma=gamma)
        with np.errstate(all="ignore"):
            r, v = p.compute_forward_posterior(ax, b)
            an, bn = p.compute_forward_message(ax, b)
        l1_cases.append(dict(gamma=gamma, ax=ax, b=b, r=r, v=v, a_new=an, b_new=bn,
                             mask=v_soft_threshold(b, gamma)))
    out["l1"] = l1_cases

    # --- Gaussian prior / likelihood
    p = GaussianPrior(size=N, mean=0.3, var=2.0)
    an, bn = p.compute_forward_message(1.3, b_rand)
    r, v = p.compute_forward_posterior(1.3, b_rand)
    out["gauss_prior"] = dict(mean=0.3, var=2.0, ax=1.3, b=b_rand, a_new=an, b_new=bn,
                              r=r, v=v, se_a=p.compute_forward_state_evolution(1.3),
                              err=p.compute_forward_error(1.3))
    yv = rs.standard_normal(N)
    lk = GaussianLikelihood(y=yv, var=0.01)
    an, bn = lk.compute_backward_message(1.3, b_rand)
    r, v = lk.compute_backward_posterior(1.3, b_rand, yv)
    out["gauss_lik"] = dict(var=0.01, az=1.3, b=b_rand, y=yv, a_new=an, b_new=bn,
                            r=r, v=v, err=lk.compute_backward_error(1.3, 1.0))

    # --- Sgn / Abs likelihoods
    z = rs.standard_normal(N)
    bz_rand = message_law(rs, 1.3, z)
    sgn_cases = []
    y_sgn = np.sign(z)
    bz_edge = np.array([0., 40., -40., 1e-9, 5., -5., 12., -12., 100., -100.])
    y_edge = np.array([1., 1., 1., -1., -1., 1., -1., 1., -1., -1.])
    for (az, bz, y) in [(1.3, kat_b, kat_y), (1.3, bz_rand, y_sgn), (0.02, bz_rand * 0.1, y_sgn),
                        (45.0, bz_rand * 30, y_sgn), (1.3, bz_edge, y_edge),
                        (1e-11, bz_rand * 1e-6, y_sgn)]:
        lk = SgnLikelihood(y=y)
        with np.errstate(all="ignore"):
            r, v = lk.compute_backward_posterior(az, bz, y)
            an, bn = lk.compute_backward_message(az, bz)
            _, vd = SgnLikelihood(y=y, isotropic=False).compute_backward_posterior(az, bz, y)
        sgn_cases.append(dict(az=az, bz=bz, y=y, r=r, v=v, a_new=an, b_new=bn, v_diag=vd))
    out["sgn_lik"] = sgn_cases
    abs_cases = []
    y_abs = np.abs(z)
    for (az, bz, y) in [(1.3, kat_b, np.abs(kat_y) * np.array([.5, 1, 2, 0, .1, 3.])),
                        (1.3, bz_rand, y_abs), (30.0, bz_rand * 20, y_abs),
                        (1.3, bz_edge, np.abs(bz_edge) * .1)]:
        lk = AbsLikelihood(y=y)
        r, v = lk.compute_backward_posterior(az, bz, y)
        an, bn = lk.compute_backward_message(az, bz)
        _, vd = AbsLikelihood(y=y, isotropic=False).compute_backward_posterior(az, bz, y)
        abs_cases.append(dict(az=az, bz=bz, y=y, r=r, v=v, a_new=an, b_new=bn, v_diag=vd))
    out["abs_lik"] = abs_cases

    # --- piecewise-linear channels
    pwl_cases = []
    chans = dict(sgn=SgnChannel, abs=AbsChannel, relu=ReluChannel)
    bx_rand = {}
    for kind, cls in chans.items():
        ch = cls()
        xk = ch.sample(z)
        bx_rand[kind] = message_law(rs, 0.7, xk)
    bz_pe = np.array([0., 30., -30., 1e-9, 5., -5., 60., -60., 0.3, -0.3, 45., -45.])
    bx_pe = np.array([0., 5., 5., -1e-9, -40., 40., 0., 0., 0., 0., 45., 45.])
    # reference tests/test_channels.py:55-59 records
    recs = [dict(az=2.1, bz=2.0, ax=2.0, bx=2.0), dict(az=2.0, bz=+1.6, ax=1.5, bx=1.3),
            dict(az=2.0, bz=-1.6, ax=1.5, bx=1.3)]
    rec_bz = np.array([r_["bz"] for r_ in recs])
    rec_bx = np.array([r_["bx"] for r_ in recs])
    for kind, cls in chans.items():
        ch = cls()
        for (az, bz, ax, bx) in [
            (1.3, kat_b, 0.7, kat_bx),
            (1.3, bz_rand, 0.7, bx_rand[kind]),
            (1.3, bz_rand, 0., np.zeros(N)),              # first iteration: ax=0, bx=0
            (1.3, bz_rand, 0, np.zeros(N)),
            (0.05, bz_rand * .05, 12.0, bx_rand[kind] * 15),
            (2.0, bz_pe, 1.5, bx_pe),
            (2.0, rec_bz, 1.5, rec_bx),
            (1e-11, bz_rand * 1e-8, 3.0, bx_rand[kind] * 3),
            (25., bz_rand * 20, 1e-11, bx_rand[kind] * 1e-9),
        ]:
            with np.errstate(all="ignore"):
                rf, vf = ch.compute_forward_posterior(az, bz, ax, bx)
                af, bf = ch.compute_forward_message(az, bz, ax, bx)
                rb, vb = ch.compute_backward_posterior(az, bz, ax, bx)
                ab, bb = ch.compute_backward_message(az, bz, ax, bx)
            pwl_cases.append(dict(kind=kind, az=float(az), bz=bz, ax=float(ax), bx=bx,
                                  r_fwd=rf, v_fwd=vf, a_fwd=af, b_fwd=bf,
                                  r_bwd=rb, v_bwd=vb, a_bwd=ab, b_bwd=bb))
    out["pwl"] = pwl_cases

    # --- section 8f rank 2: the other two-half-line channels (generic K7 form)
    pwl2_cases = []
    for kind, param, ch in [("l-relu", 0.1, LeakyReluChannel(slope=0.1)), ("l-relu", 0.5, LeakyReluChannel(slope=0.5)),
                            ("a-abs", 1e-4, AsymmetricAbsChannel(shift=1e-4)),
                            ("a-abs", 0.3, AsymmetricAbsChannel(shift=0.3))]:
        xk = ch.sample(z)
        bxk = message_law(rs, 0.7, xk)
        for (az, bz, ax, bx) in [
            (1.3, kat_b, 0.7, kat_bx),
            (1.3, bz_rand, 0.7, bxk),
            (1.3, bz_rand, 0., np.zeros(N)),
            (0.05, bz_rand * .05, 12.0, bxk * 15),
            (2.0, bz_pe, 1.5, bx_pe),
            (2.0, rec_bz, 1.5, rec_bx),
        ]:
            with np.errstate(all="ignore"):
                rf, vf = ch.compute_forward_posterior(az, bz, ax, bx)
                af, bf = ch.compute_forward_message(az, bz, ax, bx)
                rb, vb = ch.compute_backward_posterior(az, bz, ax, bx)
                ab, bb = ch.compute_backward_message(az, bz, ax, bx)
            pwl2_cases.append(dict(kind=kind, param=param, az=float(az), bz=bz, ax=float(ax), bx=bx,
                                   r_fwd=rf, v_fwd=vf, a_fwd=af, b_fwd=bf,
                                   r_bwd=rb, v_bwd=vb, a_bwd=ab, b_bwd=bb, tau_x=ch.second_moment(1.7)))
    out["pwl2"] = pwl2_cases

    # --- three-region channels (finite-interval truncated normal)
    pwl3_cases = []
    for kind, param, ch in [("h-tanh", None, HardTanhChannel()), ("h-sigm", None, HardSigmoidChannel()),
                            ("door", 0.8, SymmetricDoorChannel(width=0.8)), ("door", 0.2, SymmetricDoorChannel(width=0.2))]:
        xk = ch.sample(z)
        bxk = message_law(rs, 0.7, xk)
        for (az, bz, ax, bx) in [
            (1.3, kat_b, 0.7, kat_bx),
            (1.3, bz_rand, 0.7, bxk),
            (1.3, bz_rand, 0., np.zeros(N)),
            (0.05, bz_rand * .05, 12.0, bxk * 15),
            (2.0, bz_pe, 1.5, bx_pe),
            (2.0, rec_bz, 1.5, rec_bx),
            (30.0, bz_rand * 25, 0.7, bxk),
        ]:
            with np.errstate(all="ignore"):
                rf, vf = ch.compute_forward_posterior(az, bz, ax, bx)
                af, bf = ch.compute_forward_message(az, bz, ax, bx)
                rb, vb = ch.compute_backward_posterior(az, bz, ax, bx)
                ab, bb = ch.compute_backward_message(az, bz, ax, bx)
            pwl3_cases.append(dict(kind=kind, param=param, az=float(az), bz=bz, ax=float(ax), bx=bx,
                                   r_fwd=rf, v_fwd=vf, a_fwd=af, b_fwd=bf,
                                   r_bwd=rb, v_bwd=vb, a_bwd=ab, b_bwd=bb, tau_x=ch.second_moment(1.7)))
    out["pwl3"] = pwl3_cases

    # --- (Leaky)Relu likelihoods
    relu_lik = []
    for slope in (0.0, 0.2):
        y_obs = np.where(z >= 0, z, slope * z)
        y_kat = np.array([1.5, 0.7, 2., 0.3, 0.1, 3.]) if slope else np.array([1.5, 0., 2., 0., 0., 3.])
        if slope:
            y_kat = y_kat * np.array([1, -1, 1, -1, -1, 1.])
        for (az, bz, y) in [(1.3, kat_b, y_kat), (1.3, bz_rand, y_obs), (0.05, bz_rand * .1, y_obs),
                            (40., bz_rand * 30, y_obs)]:
            lk = ReluLikelihood(y=y) if slope == 0 else LeakyReluLikelihood(slope=slope, y=y)
            with np.errstate(all="ignore"):
                r, v = lk.compute_backward_posterior(az, bz, y)
                an, bn = lk.compute_backward_message(az, bz)
            relu_lik.append(dict(slope=slope, az=az, bz=bz, y=y, r=r, v=v, a_new=an, b_new=bn))
    out["relu_lik"] = relu_lik

    # --- log_Phi mask
    xs = np.array([-1e3, -40., -5., -1., 0., 1., 5., 10., 29.9, np.nextafter(30, 0), 30.,
                   np.nextafter(30, 31), 31., 100., 1e4])
    out["log_Phi"] = dict(x=xs, y=log_Phi(xs), mask=(xs < 30))
    save("factors", data=np.array(out, dtype=object))


# ---------------------------------------------------------------- linear
def linear_goldens():
    out = []
    for (M, N, seed) in [(12, 24, 1), (32, 64, 2), (40, 40, 3), (48, 32, 4), (100, 200, 5)]:
        rs = np.random.RandomState(seed)
        W = rs.randn(M, N) / np.sqrt(N)
        ch = LinearChannel(W)
        bz = rs.standard_normal(N) * 2
        bx = rs.standard_normal(M) * 3
        bzK = rs.standard_normal((N, 3))
        bxK = rs.standard_normal((M, 3))
        cases = []
        for (az, ax) in [(1.3, 0.7), (2.0, 0.), (2.0, 0), (1e-11, 5.0), (1e11, 1e-11),
                         (100.0, 100.0), (0.37, 1e4)]:
            rz = ch.compute_backward_mean(az, bz, ax, bx)
            rx = ch.compute_forward_mean(az, bz, ax, bx)
            rzK = ch.compute_backward_mean(az, bzK, ax, bxK)
            vz = ch.compute_backward_variance(az, ax)
            vx = ch.compute_forward_variance(az, ax)
            af, bf = ch.compute_forward_message(az, bz, ax, bx)
            ab, bb = ch.compute_backward_message(az, bz, ax, bx)
            cases.append(dict(az=float(az), ax=float(ax), rz=rz, rx=rx, rzK=rzK, vz=vz, vx=vx,
                              a_fwd=af, b_fwd=bf, a_bwd=ab, b_bwd=bb,
                              n_eff=ch.compute_n_eff(az, ax)))
        # az = 0 only enters the variance path (linear_channel.py:92-94)
        vz0 = ch.compute_backward_variance(0., 0.7)
        out.append(dict(M=M, N=N, seed=seed, W_sum=W.sum(), W_abs_sum=np.abs(W).sum(),
                        bz=bz, bx=bx, bzK=bzK, bxK=bxK, cases=cases, vz_az0=vz0,
                        rank=ch.rank, spectrum=ch.spectrum,
                        second_moment=ch.second_moment(1.7)))
    # rank-deficient W (rank < min(M, N)): singular = spectrum[:rank]
    rs = np.random.RandomState(6)
    W = rs.randn(20, 6) @ rs.randn(6, 30) / 30
    ch = LinearChannel(W)
    bz = rs.standard_normal(30)
    bx = rs.standard_normal(20)
    out.append(dict(M=20, N=30, seed=6, lowrank=6, W=W, bz=bz, bx=bx, rank=ch.rank,
                    spectrum=ch.spectrum,
                    cases=[dict(az=1.3, ax=0.7, rz=ch.compute_backward_mean(1.3, bz, 0.7, bx),
                                rx=ch.compute_forward_mean(1.3, bz, 0.7, bx),
                                vz=ch.compute_backward_variance(1.3, 0.7),
                                vx=ch.compute_forward_variance(1.3, 0.7),
                                n_eff=ch.compute_n_eff(1.3, 0.7))]))
    save("linear", data=np.array(out, dtype=object))


# ---------------------------------------------------------------- EP trajectories
class Recorder:
    "Callback capturing every edge's (a, b) and every variable's (r, v)."

    def __init__(self, stopper):
        self.stopper = stopper
        self.iters = []

    def __call__(self, algo, i, max_iter):
        if i == 0:
            self.iters = []
        rec = {}
        for source, target, data in algo.message_dag.edges(data=True):
            if isinstance(target, Variable):          # factor -> variable edges
                key = f"{data['direction']}:{target.id}"
                rec[key + ":a"] = float(data["a"])
                if "b" in data:
                    rec[key + ":b"] = np.array(data["b"], dtype=float)
        for vid, d in algo.get_variables_data().items():
            if "r" in d:
                rec[f"r:{vid}"] = np.array(d["r"], dtype=float)
            rec[f"v:{vid}"] = float(d["v"])
        self.iters.append(rec)
        return self.stopper(algo, i, max_iter)


def pack_traj(iters):
    keys = iters[0].keys()
    return {k: np.stack([np.asarray(it[k]) for it in iters]) for k in keys}


def teacher_glm(seed, N, alpha, rho, output, noise_var):
    rs = np.random.RandomState(seed)
    M = int(alpha * N)
    W = rs.randn(M, N) / np.sqrt(N)
    x = rs.standard_normal(N) * rs.binomial(n=1, size=N, p=rho)
    z = W @ x
    if output == "gaussian":
        y = z + np.sqrt(noise_var) * rs.standard_normal(M)
    elif output == "sgn":
        y = np.sign(z)
    elif output == "abs":
        y = np.abs(z)
    return W, x, y


def ep_goldens():
    trajs = []
    for output, alpha in [("gaussian", 0.5), ("sgn", 0.7), ("abs", 1.5)]:
        for N in (64, 256):
            for seed in (1, 2, 3):
                for damping in (None, 0.1):
                    if damping and seed != 1:
                        continue
                    rho, nv = (0.05, 1e-2) if output == "gaussian" else (0.25, None)
                    if N == 64:
                        rho = 0.25
                    W, x, y = teacher_glm(seed, N, alpha, rho, output, nv)
                    prior = GaussBernoulliPrior(size=N, rho=rho)
                    lin = LinearChannel(W)
                    if output == "gaussian":
                        lik = GaussianLikelihood(y=y, var=nv)
                    elif output == "sgn":
                        lik = SgnLikelihood(y=y)
                    else:
                        lik = AbsLikelihood(y=y)
                    model = (prior @ V(id="x") @ lin @ V(id="z") @ lik).to_model()
                    ep = ExpectationPropagation(model)
                    rec = Recorder(EarlyStoppingEP())
                    err = None
                    try:
                        with np.errstate(all="ignore"):
                            ep.iterate(max_iter=30, callback=rec, damping=damping)
                    except Exception as e:    # noqa: BLE001
                        err = repr(e)
                    t = pack_traj(rec.iters[:12])
                    trajs.append(dict(kind="glm", output=output, N=N, alpha=alpha, seed=seed,
                                      rho=rho, noise_var=nv, damping=damping, y=y, x=x,
                                      W_sum=W.sum(), n_iter=ep.n_iter, error=err,
                                      n_recorded=len(rec.iters), traj=t))
                    print("glm", output, N, seed, damping, "n_iter", ep.n_iter, err)
    # two-layer:  Gaussian prior -> Linear(W1) -> Relu -> Linear(W2) -> Gaussian likelihood
    for N in (64, 200):
        for seed in (1, 2):
            rs = np.random.RandomState(seed)
            M1, M2 = N // 2 * 2, N // 2
            if N == 64:
                M1 = 128       # expanding first layer, as the reference's alpha=2 example
            W1 = rs.randn(M1, N) / np.sqrt(N)
            W2 = rs.randn(M2, M1) / np.sqrt(M1)
            x = rs.standard_normal(N)
            h = np.maximum(0, W1 @ x)
            nv = 1e-2
            y = W2 @ h + np.sqrt(nv) * rs.standard_normal(M2)
            model = (GaussianPrior(size=N) @ V(id="x") @ LinearChannel(W1) @ V(id="z1")
                     @ ReluChannel() @ V(id="a1") @ LinearChannel(W2) @ V(id="z2")
                     @ GaussianLikelihood(y=y, var=nv)).to_model()
            ep = ExpectationPropagation(model)
            rec = Recorder(EarlyStoppingEP())
            err = None
            try:
                with np.errstate(all="ignore"):
                    ep.iterate(max_iter=30, callback=rec, damping=None)
            except Exception as e:    # noqa: BLE001
                err = repr(e)
            t = pack_traj(rec.iters[:12])
            trajs.append(dict(kind="two_layer", N=N, M1=M1, M2=M2, seed=seed, noise_var=nv,
                              y=y, x=x, W1_sum=W1.sum(), W2_sum=W2.sum(), n_iter=ep.n_iter,
                              error=err, n_recorded=len(rec.iters), traj=t))
            print("two_layer", N, seed, "n_iter", ep.n_iter, err)
    save("ep_traj", data=np.array(trajs, dtype=object))


# ---------------------------------------------------------------- three-region likelihoods
def pwl3_lik_goldens():
    "HardTanh / HardSigmoid / SymmetricDoor likelihoods: posteriors, messages, SE errors, EP runs."
    rs = np.random.RandomState(21)
    n = 96
    mk = {"h-tanh": lambda y, p: HardTanhLikelihood(y=y), "h-sigm": lambda y, p: HardSigmoidLikelihood(y=y),
          "door": lambda y, p: SymmetricDoorLikelihood(width=p, y=y)}
    cases = []
    for kind, param in (("h-tanh", None), ("h-sigm", None), ("door", 0.7), ("door", 1.5)):
        z = rs.standard_normal(n) * 2.5
        y = mk[kind](None, param).sample(z)
        for (az, scale) in ((1.3, 1.0), (0.05, 0.1), (40., 30.), (3.0, 4.0)):
            bz = az * z + np.sqrt(az) * rs.standard_normal(n) * scale / max(scale, 1.0)
            lk = mk[kind](y, param)
            with np.errstate(all="ignore"):
                r, v = lk.compute_backward_posterior(az, bz, y)
                an, bn = lk.compute_backward_message(az, bz)
            cases.append(dict(kind=kind, param=param, az=az, bz=bz, y=y, r=r, v=v, a_new=an, b_new=bn))
    se = []
    for kind, param in (("h-tanh", None), ("h-sigm", None), ("door", 0.7)):
        lk = mk[kind](None, param)
        for (az, tau_z) in ((1.3, 1.0), (3.0, 0.5), (25., 2.0)):
            t0 = time.time()
            se.append(dict(kind=kind, param=param, az=az, tau_z=tau_z, v=lk.compute_backward_error(az, tau_z)))
            print("se pwl3 lik", kind, az, tau_z, se[-1]["v"], f"{time.time()-t0:.1f}s")
    trajs = []
    for kind, param, alpha in (("door", 0.7, 2.0), ("h-tanh", None, 1.0)):
        N = 128
        M = int(alpha * N)
        r = np.random.RandomState(3)
        W = r.randn(M, N) / np.sqrt(N)
        x = r.standard_normal(N)
        y = mk[kind](None, param).sample(W @ x)
        model = (GaussianPrior(size=N) @ V(id="x") @ LinearChannel(W) @ V(id="z") @ mk[kind](y, param)).to_model()
        ep = ExpectationPropagation(model)
        rec = Recorder(EarlyStoppingEP())
        err = None
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=30, callback=rec, damping=0.2)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        trajs.append(dict(kind=kind, param=param, N=N, M=M, seed=3, damping=0.2, x=x, y=y, W_sum=W.sum(),
                          n_iter=ep.n_iter, error=err, n_recorded=len(rec.iters), traj=pack_traj(rec.iters[:12])))
        print("pwl3 lik ep", kind, "n_iter", ep.n_iter, err)
    save("pwl3_lik", data=np.array(cases, dtype=object), se=np.array(se, dtype=object),
         ep=np.array(trajs, dtype=object))


# ---------------------------------------------------------------- unitary channels (DFT, rotation)
def unitary_goldens():
    rs = np.random.RandomState(31)
    msgs = []
    for shape in ((16,), (15,), (6, 8)):
        ch = DFTChannel(real=True)
        bz = rs.standard_normal(shape)
        bx = rs.standard_normal((2,) + shape)
        af, bf = ch.compute_forward_message(1.3, bz, 0.7, bx)
        ab, bb = ch.compute_backward_message(1.3, bz, 0.7, bx)
        msgs.append(dict(kind="dft", shape=shape, bz=bz, bx=bx, a_fwd=af, b_fwd=bf, a_bwd=ab, b_bwd=bb,
                         x=ch.sample(bz)))
    for N in (12, 33):
        Q, _ = np.linalg.qr(rs.randn(N, N))
        ch = RotationChannel(Q)
        bz, bx = rs.standard_normal(N), rs.standard_normal(N)
        af, bf = ch.compute_forward_message(1.3, bz, 0.7, bx)
        ab, bb = ch.compute_backward_message(1.3, bz, 0.7, bx)
        msgs.append(dict(kind="rotation", R=Q, bz=bz, bx=bx, a_fwd=af, b_fwd=bf, a_bwd=ab, b_bwd=bb))
    trajs = []
    for kind, damping in (("dft", None), ("dft", 0.3), ("rotation", None), ("rotation", 0.3)):
        r = np.random.RandomState(7)
        N, rho, nv = 96, 0.15, 1e-2
        x = r.standard_normal(N) * r.binomial(n=1, size=N, p=rho)
        if kind == "dft":
            ch, Q = DFTChannel(real=True), None
        else:
            Q, _ = np.linalg.qr(r.randn(N, N))
            ch = RotationChannel(Q)
        z = ch.sample(x)
        y = z + np.sqrt(nv) * r.standard_normal(z.shape)
        model = (GaussBernoulliPrior(size=N, rho=rho) @ V(id="x") @ ch @ V(id="z")
                 @ GaussianLikelihood(y=y, var=nv)).to_model()
        ep = ExpectationPropagation(model)
        rec = Recorder(EarlyStoppingEP())
        err = None
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=30, callback=rec, damping=damping)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        trajs.append(dict(kind=kind, N=N, rho=rho, noise_var=nv, damping=damping, x=x, y=y, R=Q, n_iter=ep.n_iter,
                          error=err, n_recorded=len(rec.iters), traj=pack_traj(rec.iters[:12])))
        print("unitary ep", kind, damping, "n_iter", ep.n_iter, err)
    save("unitary", data=np.array(msgs, dtype=object), ep=np.array(trajs, dtype=object))


# ---------------------------------------------------------------- log-partitions, log-evidence
def logz_goldens():
    "compute_log_partition of the hot-path factors and the EP log-evidence along a run."
    rs = np.random.RandomState(41)
    n = 80
    bx = rs.standard_normal(n) * 2
    bz = rs.standard_normal(n) * 1.5
    y_sgn = np.sign(rs.standard_normal(n))
    y_abs = np.abs(rs.standard_normal(n))
    y_g = rs.standard_normal(n)
    out = []
    for (az, ax) in [(1.3, 0.7), (0.05, 4.0), (40., 0.2)]:
        c = dict(az=az, ax=ax, bz=bz, bx=bx, y_sgn=y_sgn, y_abs=y_abs, y_g=y_g)
        c["gb"] = GaussBernoulliPrior(size=n, rho=0.1, mean=0.3, var=1.7).compute_log_partition(ax, bx)
        c["gauss_prior"] = GaussianPrior(size=n, mean=0.3, var=1.7).compute_log_partition(ax, bx)
        c["l1"] = MAP_L1NormPrior(size=n, gamma=0.8).compute_log_partition(ax, bx)
        c["gauss_lik"] = GaussianLikelihood(y=y_g, var=0.4).compute_log_partition(az, bz, y_g)
        c["sgn_lik"] = SgnLikelihood(y=y_sgn).compute_log_partition(az, bz, y_sgn)
        c["abs_lik"] = AbsLikelihood(y=y_abs).compute_log_partition(az, bz, y_abs)
        for name, ch in (("sgn", SgnChannel()), ("abs", AbsChannel()), ("relu", ReluChannel())):
            c["pwl_" + name] = ch.compute_log_partition(az, bz, ax, bx)
        c["variable"] = V(id="v").compute_log_partition(az + ax, bz + bx)
        out.append(c)
    lin = []
    for (M, N, seed) in [(12, 24, 1), (40, 40, 3), (48, 32, 4)]:
        r = np.random.RandomState(seed)
        W = r.randn(M, N) / np.sqrt(N)
        ch = LinearChannel(W)
        b_z, b_x = r.standard_normal(N) * 2, r.standard_normal(M) * 3
        lin.append(dict(M=M, N=N, seed=seed, W=W, bz=b_z, bx=b_x,
                        cases=[dict(az=az, ax=ax, logZ=ch.compute_log_partition(az, b_z, ax, b_x))
                               for (az, ax) in [(1.3, 0.7), (0.2, 5.0), (30., 0.1)]]))
    # log-evidence along EP runs
    runs = []
    for output, alpha in [("gaussian", 0.5), ("sgn", 0.7), ("abs", 1.5)]:
        N, seed, damping = 64, 1, 0.1
        rho, nv = 0.25, (1e-2 if output == "gaussian" else None)
        W, x, y = teacher_glm(seed, N, alpha, rho, output, nv)
        lik = (GaussianLikelihood(y=y, var=nv) if output == "gaussian"
               else SgnLikelihood(y=y) if output == "sgn" else AbsLikelihood(y=y))
        model = (GaussBernoulliPrior(size=N, rho=rho) @ V(id="x") @ LinearChannel(W) @ V(id="z") @ lik).to_model()
        ep = ExpectationPropagation(model)
        As = []

        def cb(algo, i, max_iter, As=As):
            As.append(float(algo.log_evidence()))
        with np.errstate(all="ignore"):
            ep.iterate(max_iter=10, callback=cb, damping=damping)
        runs.append(dict(kind="glm", output=output, N=N, alpha=alpha, seed=seed, rho=rho, noise_var=nv,
                         damping=damping, y=y, W_sum=W.sum(), A=np.array(As)))
        print("logz glm", output, As[:3], As[-1])
    rs2 = np.random.RandomState(2)
    N, M1, M2 = 48, 64, 24
    W1 = rs2.randn(M1, N) / np.sqrt(N)
    W2 = rs2.randn(M2, M1) / np.sqrt(M1)
    x = rs2.standard_normal(N)
    y = W2 @ np.maximum(0, W1 @ x) + 0.1 * rs2.standard_normal(M2)
    model = (GaussianPrior(size=N) @ V(id="x") @ LinearChannel(W1) @ V(id="z1") @ ReluChannel() @ V(id="a1")
             @ LinearChannel(W2) @ V(id="z2") @ GaussianLikelihood(y=y, var=1e-2)).to_model()
    ep = ExpectationPropagation(model)
    As = []

    def cb2(algo, i, max_iter):
        As.append(float(algo.log_evidence()))
    with np.errstate(all="ignore"):
        ep.iterate(max_iter=8, callback=cb2, damping=0.1)
    runs.append(dict(kind="two_layer", N=N, M1=M1, M2=M2, seed=2, noise_var=1e-2, damping=0.1, y=y,
                     W1=W1, W2=W2, A=np.array(As)))
    print("logz two_layer", As[:3], As[-1])
    save("logz", data=np.array(out, dtype=object), lin=np.array(lin, dtype=object), ep=np.array(runs, dtype=object))


# ---------------------------------------------------------------- convolution channel
def conv_goldens():
    "ConvChannel family (real=True): posteriors, messages, scalars, samples, EP trajectories."
    rs = np.random.RandomState(11)
    chans = [
        ("conv", dict(filter=rs.randn(16) / 4)),
        ("conv", dict(filter=rs.randn(15) / 4)),
        ("conv", dict(filter=rs.randn(6, 8) / 7)),
        ("conv", dict(filter=rs.randn(3, 4, 5) / 8)),
        ("blur_1d", dict(sigma=2.0, N=32)),
        ("blur_2d", dict(sigma=[1.0, 2.0], shape=(8, 10))),
        ("laplacian", dict(shape=(8, 8))),
        ("diff", dict(D1=[1.0, -0.5, 0.25], D2=[0.3, 0.0, -0.2], shape=(3, 4, 5))),
        ("diff", dict(D1=[1.0], D2=None, shape=(9,))),
    ]
    ctor = dict(conv=ConvChannel, blur_1d=Blur1DChannel, blur_2d=Blur2DChannel,
                laplacian=LaplacianChannel, diff=DifferentialChannel)
    out = []
    for kind, kw in chans:
        ch = ctor[kind](**kw)
        shape = ch.shape
        bz = rs.standard_normal(shape) * 2
        bx = rs.standard_normal(shape) * 3
        z = rs.standard_normal(shape)
        cases = []
        for (az, ax) in [(1.3, 0.7), (2.0, 0.), (1e-11, 5.0), (1e11, 1e-11), (100.0, 100.0),
                         (0.37, 1e4), (1e-300, 3.0)]:
            with np.errstate(all="ignore"):
                rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
                rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
                af, bf = ch.compute_forward_message(az, bz, ax, bx)
                ab, bb = ch.compute_backward_message(az, bz, ax, bx)
            cases.append(dict(az=float(az), ax=float(ax), rz=rz, rx=rx, vz=vz, vx=vx, a_fwd=af, b_fwd=bf,
                              a_bwd=ab, b_bwd=bb, n_eff=ch.compute_n_eff(az, ax)))
        out.append(dict(kind=kind, kwargs=kw, filter=np.array(ch.filter), bz=bz, bx=bx, z=z,
                        x=ch.sample(z), spectrum=ch.spectrum, second_moment=ch.second_moment(1.7),
                        cases=cases))
    # EP: sparse deconvolution  GaussBernoulli -> Conv -> Gaussian likelihood
    trajs = []
    for name, shape, make, rho, damping in [
            ("conv1d", (128,), lambda r: ConvChannel(filter=r.randn(128) / np.sqrt(128)), 0.1, None),
            ("conv1d", (128,), lambda r: ConvChannel(filter=r.randn(128) / np.sqrt(128)), 0.1, 0.2),
            ("conv1d_odd", (99,), lambda r: ConvChannel(filter=r.randn(99) / np.sqrt(99)), 0.2, None),
            ("blur2d", (12, 16), lambda r: Blur2DChannel(sigma=[0.6, 0.8], shape=(12, 16)), 0.15, None)]:
        r = np.random.RandomState(5)
        ch = make(r)
        x = r.standard_normal(shape) * r.binomial(n=1, size=shape, p=rho)
        nv = 1e-3
        y = ch.sample(x) + np.sqrt(nv) * r.standard_normal(shape)
        model = (GaussBernoulliPrior(size=shape, rho=rho) @ V(id="x") @ ch @ V(id="z")
                 @ GaussianLikelihood(y=y, var=nv)).to_model()
        ep = ExpectationPropagation(model)
        rec = Recorder(EarlyStoppingEP())
        err = None
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=40, callback=rec, damping=damping)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        trajs.append(dict(name=name, shape=shape, rho=rho, noise_var=nv, damping=damping, x=x, y=y,
                          filter=np.array(ch.filter), n_iter=ep.n_iter, error=err,
                          n_recorded=len(rec.iters), traj=pack_traj(rec.iters[:12]),
                          final_r=rec.iters[-1]["r:x"], mse=float(np.mean((rec.iters[-1]["r:x"] - x)**2))))
        print("conv ep", name, damping, "n_iter", ep.n_iter, err, "mse", trajs[-1]["mse"])
    save("conv", data=np.array(out, dtype=object), ep=np.array(trajs, dtype=object))


# ---------------------------------------------------------------- trees: gradient channel, multi-factor variables
class TreeRecorder:
    "Every message of both kinds (factor -> variable, variable -> factor), keyed by class name / variable id."

    def __init__(self, stopper):
        self.stopper = stopper
        self.iters = []

    @staticmethod
    def name(node):
        return node.id if isinstance(node, Variable) else type(node).__name__

    def __call__(self, algo, i, max_iter):
        if i == 0:
            self.iters = []
        rec = {}
        for source, target, data in algo.message_dag.edges(data=True):
            key = f"{self.name(source)}->{self.name(target)}"
            rec[key + ":a"] = float(data["a"])
            rec[key + ":b"] = np.array(data["b"], dtype=float)
        for vid, d in algo.get_variables_data().items():
            rec[f"r:{vid}"] = np.array(d["r"], dtype=float)
            rec[f"v:{vid}"] = float(d["v"])
        self.iters.append(rec)
        return self.stopper(algo, i, max_iter)


def tree_goldens():
    "GradientChannel / UnitaryChannel factor-level vectors and EP on the sparse-gradient tree."
    from tramp.channels import GradientChannel, UnitaryChannel
    from tramp.variables import SIMOVariable, MILeafVariable
    rs = np.random.RandomState(23)
    grads = []
    for shape in [(16,), (15,), (6, 8), (3, 4, 5)]:
        ch = GradientChannel(shape=shape)
        d = len(shape)
        bz = rs.standard_normal(shape) * 2
        bx = rs.standard_normal((d,) + shape) * 3
        z = rs.standard_normal(shape)
        cases = []
        for (az, ax) in [(1.3, 0.7), (2.0, 0.), (1e-11, 5.0), (100.0, 100.0), (0.37, 1e4)]:
            with np.errstate(all="ignore"):
                rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
                rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
                af, bf = ch.compute_forward_message(az, bz, ax, bx)
                ab, bb = ch.compute_backward_message(az, bz, ax, bx)
            cases.append(dict(az=float(az), ax=float(ax), rz=rz, rx=rx, vz=vz, vx=vx, a_fwd=af, b_fwd=bf,
                              a_bwd=ab, b_bwd=bb, n_eff=ch.compute_n_eff(az, ax)))
        grads.append(dict(shape=shape, bz=bz, bx=bx, z=z, x=ch.sample(z), spectrum=ch.spectrum,
                          second_moment=ch.second_moment(1.7), cases=cases))
    unit = []
    for N in (5, 12):
        A = rs.standard_normal((N, N)) + 1j * rs.standard_normal((N, N))
        U, _ = np.linalg.qr(A)
        ch = UnitaryChannel(U)
        bz = rs.standard_normal((2, N))
        bx = rs.standard_normal((2, N))
        af, bf = ch.compute_forward_message(1.3, bz, 0.7, bx)
        ab, bb = ch.compute_backward_message(1.3, bz, 0.7, bx)
        unit.append(dict(U=U, bz=bz, bx=bx, a_fwd=af, b_fwd=bf, a_bwd=ab, b_bwd=bb, x=ch.sample(bz)))
    # EP on the sparse-gradient model (examples/sparse/plot_sparse_gradient.py, models/total_variation_model.py):
    # x ~ N(0, 1) observed through Gaussian noise, z = grad x with a Gauss-Bernoulli prior on z
    trajs = []
    for name, shape, rho, nv, damping in [("grad1d", (64,), 0.06, 1e-2, None), ("grad1d", (64,), 0.06, 1e-2, 0.3),
                                          ("grad2d", (8, 10), 0.15, 1e-2, None)]:
        r = np.random.RandomState(9)
        d = len(shape)
        zshape = (d,) + shape
        if d == 1:
            zp = r.standard_normal(shape) * r.binomial(n=1, size=shape, p=rho)
            x = np.cumsum(zp)
            x = x - x.mean()
        else:
            x = np.kron(r.standard_normal((2, 2)), np.ones((shape[0] // 2, shape[1] // 2)))
        y = x + np.sqrt(nv) * r.standard_normal(shape)
        model = (GaussianPrior(size=shape) @ SIMOVariable(id="x", n_next=2) @ (
            GaussianLikelihood(y=y, var=nv) + (
                GradientChannel(shape=shape) + GaussBernoulliPrior(size=zshape, rho=rho)
            ) @ MILeafVariable(id="z", n_prev=2))).to_model()
        ep = ExpectationPropagation(model)
        rec = TreeRecorder(EarlyStoppingEP())
        err = None
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=30, callback=rec, damping=damping)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        trajs.append(dict(name=name, shape=shape, rho=rho, noise_var=nv, damping=damping, x=x, y=y,
                          n_iter=ep.n_iter, error=err, n_recorded=len(rec.iters), traj=pack_traj(rec.iters[:12]),
                          final_r=rec.iters[-1]["r:x"], mse=float(np.mean((rec.iters[-1]["r:x"] - x)**2)),
                          mse_y=float(np.mean((y - x)**2))))
        print("tree ep", name, damping, "n_iter", ep.n_iter, err, "mse", trajs[-1]["mse"], "vs y", trajs[-1]["mse_y"])
    # sparse-gradient regression (models/total_variation_model.py:12-25,43-48,67-71, built by hand: the module
    # itself calls np.product, gone from NumPy 2): the tree above feeding a dense linear measurement
    from tramp.channels import ReshapeChannel
    regs = []
    for name, shape, M, rho, nv, damping in [("gradreg1d", (48,), 30, 0.08, 1e-2, None),
                                             ("gradreg1d", (48,), 30, 0.08, 1e-2, 0.2),
                                             ("gradreg2d", (6, 8), 30, 0.2, 1e-2, 0.1)]:
        r = np.random.RandomState(13)
        d, N = len(shape), int(np.prod(shape))
        if d == 1:
            x = np.cumsum(r.standard_normal(shape) * r.binomial(n=1, size=shape, p=rho))
            x = x - x.mean()
        else:
            x = np.kron(r.standard_normal((2, 2)), np.ones((shape[0] // 2, shape[1] // 2)))
        A = r.standard_normal((M, N)) / np.sqrt(N)
        y = A @ x.ravel() + np.sqrt(nv) * r.standard_normal(M)
        model = (GaussianPrior(size=shape, var=1.0) @ SIMOVariable(id="x", n_next=2) @ (
            (GradientChannel(shape=shape) + GaussBernoulliPrior(size=(d,) + shape, rho=rho))
            @ MILeafVariable(id="x'", n_prev=2)) @ ReshapeChannel(prev_shape=shape, next_shape=N)
            @ V(id="r") @ LinearChannel(A, name="A") @ V(id="z") @ GaussianLikelihood(y, var=nv)).to_model()
        ep = ExpectationPropagation(model)
        rec = TreeRecorder(EarlyStoppingEP())
        err = None
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=30, callback=rec, damping=damping)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        regs.append(dict(name=name, shape=shape, M=M, rho=rho, noise_var=nv, damping=damping, x=x, y=y, A=A,
                         n_iter=ep.n_iter, error=err, n_recorded=len(rec.iters), traj=pack_traj(rec.iters[:10]),
                         mse=float(np.mean((rec.iters[-1]["r:x"] - x)**2))))
        print("tree reg", name, damping, "n_iter", ep.n_iter, err, "mse", regs[-1]["mse"], "var x", x.var())
    # GaussianChannel as a hidden factor (channels/noise/gaussian_channel.py:23-33): noisy 1-bit compressed
    # sensing, the classification block of total_variation_model.py:51-56
    from tramp.channels import GaussianChannel
    gch = []
    for seed, N, alpha, var, damping in [(1, 64, 1.5, 0.05, None), (2, 128, 2.0, 0.2, 0.2)]:
        W, x, _ = teacher_glm(seed, N, alpha, 0.3, "gaussian", 0.0)
        r = np.random.RandomState(seed + 50)
        y = np.sign(W @ x + np.sqrt(var) * r.standard_normal(W.shape[0]))
        ch = GaussianChannel(var=var)
        bz = r.standard_normal(7)
        msgs = dict(bz=bz, fwd=ch.compute_forward_message(1.3, bz, 0.7, 2 * bz),
                    bwd=ch.compute_backward_message(1.3, bz, 0.7, 2 * bz))
        model = (GaussBernoulliPrior(size=N, rho=0.3) @ V(id="x") @ LinearChannel(W) @ V(id="z")
                 @ GaussianChannel(var=var) @ V(id="a") @ SgnLikelihood(y=y)).to_model()
        ep = ExpectationPropagation(model)
        rec = Recorder(EarlyStoppingEP())
        with np.errstate(all="ignore"):
            ep.iterate(max_iter=40, callback=rec, damping=damping)
        gch.append(dict(seed=seed, N=N, alpha=alpha, var=var, damping=damping, x=x, y=y, msgs=msgs,   # W = regen_W(seed)
                        n_iter=ep.n_iter, traj=pack_traj(rec.iters[:10]), final_r=rec.iters[-1]["r:x"]))
        print("gauss channel ep", N, damping, "n_iter", ep.n_iter)
    # state evolution on the sparse-gradient tree (observed form: a likelihood with y=None in place of the
    # noise channel + leaf, as glm_state_evolution does)
    ses = []
    for shape, rho, nv, damping in [((64,), 0.1, 1e-2, None), ((8, 10), 0.2, 5e-2, 0.2)]:
        d = len(shape)
        model = (GaussianPrior(size=shape) @ SIMOVariable(id="x", n_next=2) @ (
            GaussianLikelihood(y=None, var=nv) + (
                GradientChannel(shape=shape) + GaussBernoulliPrior(size=(d,) + shape, rho=rho)
            ) @ MILeafVariable(id="z", n_prev=2))).to_model()
        se = StateEvolution(model)
        iters = []

        def cb(algo, i, max_iter):
            rec = {}
            for source, target, data in algo.message_dag.edges(data=True):
                rec[f"{TreeRecorder.name(source)}->{TreeRecorder.name(target)}:a"] = float(data["a"])
            for vid, dd in algo.get_variables_data().items():
                rec[f"v:{vid}"] = float(dd["v"])
            iters.append(rec)
            return False
        se.iterate(max_iter=12, callback=cb, damping=damping)
        ses.append(dict(shape=shape, rho=rho, noise_var=nv, damping=damping, traj=pack_traj(iters),
                        tau=model.get_second_moments()))
        print("tree se", shape, damping, {k: v for k, v in iters[-1].items() if k.startswith("v:")})
    # MAP_L21NormPrior (axis 0) and total-variation denoising (models/total_variation_model.py:28-40):
    # the sparse-gradient tree with the group penalty on the gradient field instead of the Gauss-Bernoulli prior
    from tramp.priors import MAP_L21NormPrior
    l21 = []
    for size, gamma in [((2, 9), 1.0), ((3, 4, 5), 0.7), ((1, 12), 2.0)]:
        pr = MAP_L21NormPrior(size=size, gamma=gamma)
        bx = rs.standard_normal(size) * 1.5
        for ax in (1.3, 0.2):
            r, v = pr.compute_forward_posterior(ax, bx)
            a_new, b_new = pr.compute_forward_message(ax, bx)
            l21.append(dict(size=size, gamma=gamma, ax=ax, bx=bx, r=r, v=v, a_new=a_new, b_new=b_new,
                            mask=np.linalg.norm(bx, axis=0) > gamma))
    tvs = []
    for name, shape, gamma, nv, damping, b0 in [("tv1d", (64,), 0.5, 1e-2, 0.3, 1.0), ("tv2d", (8, 10), 0.5, 1e-2, 0.3, 1.0)]:
        r = np.random.RandomState(31)
        d = len(shape)
        if d == 1:
            x = np.cumsum(r.standard_normal(shape) * r.binomial(n=1, size=shape, p=0.08))
            x = x - x.mean()
        else:
            x = np.kron(r.standard_normal((2, 2)), np.ones((shape[0] // 2, shape[1] // 2)))
        y = x + np.sqrt(nv) * r.standard_normal(shape)
        model = (GaussianPrior(size=shape) @ SIMOVariable(id="x", n_next=2) @ (
            GaussianLikelihood(y=y, var=nv) + (
                GradientChannel(shape=shape) + MAP_L21NormPrior(size=(d,) + shape, gamma=gamma)
            ) @ MILeafVariable(id="z", n_prev=2))).to_model()
        ep = ExpectationPropagation(model)
        rec = TreeRecorder(EarlyStoppingEP())
        err = None
        # b = 0 makes every group norm 0 and the reference's variance 0 * nan: start from a constant b
        init = ConstantInit(a=0.1, b=b0)
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=30, callback=rec, damping=damping, initializer=init)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        tvs.append(dict(name=name, shape=shape, gamma=gamma, noise_var=nv, damping=damping, a0=0.1, b0=b0, x=x, y=y,
                        n_iter=ep.n_iter, error=err, traj=pack_traj(rec.iters[:10]) if rec.iters else None,
                        mse=float(np.mean((rec.iters[-1]["r:x"] - x)**2)) if rec.iters else None))
        print("tv ep", name, "n_iter", ep.n_iter, err, "mse", tvs[-1]["mse"], "y mse", float(np.mean((y - x)**2)))
    # generative (VAE-like) prior as in examples/vae_prior/plot_vae.py: Gaussian latent -> Linear -> Bias ->
    # LeakyRelu -> Linear -> Bias -> HardTanh, observed through Gaussian noise (denoising)
    from tramp.channels import BiasChannel
    vae = []
    for seed, (D, H, N), nv, damping in [(3, (8, 24, 32), 0.05, 0.3), (4, (6, 20, 40), 0.1, 0.5)]:
        r = np.random.RandomState(seed)
        W1, b1 = r.standard_normal((H, D)) / np.sqrt(D), 0.3 * r.standard_normal(H)
        W2, b2 = r.standard_normal((N, H)) / np.sqrt(H), 0.3 * r.standard_normal(N)
        z0 = r.standard_normal(D)
        h = W1 @ z0 + b1
        h = np.where(h > 0, h, 0.1 * h)
        x = np.clip(W2 @ h + b2, -1, 1)
        y = x + np.sqrt(nv) * r.standard_normal(N)
        bch = BiasChannel(b1)
        bb = r.standard_normal(H)
        msgs = dict(b=bb, fwd=bch.compute_forward_message(1.3, bb, 0.7, 2 * bb),
                    bwd=bch.compute_backward_message(1.3, bb, 0.7, 2 * bb), tau=bch.second_moment(1.7))
        model = (GaussianPrior(size=D) @ V(id="z") @ LinearChannel(W1, name="W1") @ V(id="a1") @ BiasChannel(b1)
                 @ V(id="a2") @ LeakyReluChannel(slope=0.1) @ V(id="a3") @ LinearChannel(W2, name="W2") @ V(id="a4")
                 @ BiasChannel(b2) @ V(id="a5") @ HardTanhChannel() @ V(id="x")
                 @ GaussianLikelihood(y=y, var=nv)).to_model()
        ep = ExpectationPropagation(model)
        rec = Recorder(EarlyStoppingEP())
        err = None
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=40, callback=rec, damping=damping)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        vae.append(dict(seed=seed, dims=(D, H, N), noise_var=nv, damping=damping, W1=W1, b1=b1, W2=W2, b2=b2, x=x, y=y,
                        msgs=msgs, n_iter=ep.n_iter, error=err, traj=pack_traj(rec.iters[:10]),
                        mse=float(np.mean((rec.iters[-1]["r:x"] - x)**2))))
        print("vae prior ep", (D, H, N), damping, "n_iter", ep.n_iter, err, "mse", vae[-1]["mse"], "y mse", float(np.mean((y - x)**2)))
    # committee machine (models/committee_model.py:8-41): K experts prior -> x_k -> F -> z_k -> relu -> a_k sharing
    # the sensing matrix, summed by the K-input SumChannel, observed through Gaussian noise
    from tramp.channels import SumChannel

    def nb_name(algo, node):
        "variables by id; factors by class and the ids of the variables around them (several share a class)"
        if isinstance(node, Variable):
            return node.id
        g = algo.model_dag
        ids = sorted(v.id for v in list(g.predecessors(node)) + list(g.successors(node)))
        return f"{type(node).__name__}[{','.join(ids)}]"

    com = []
    for K, N, alpha, pvars, nv, damping in [(2, 24, 2.0, [1.0, 0.5], 1e-2, 0.2), (3, 16, 3.0, [1.0, 0.6, 0.3], 5e-2, 0.3)]:
        r = np.random.RandomState(17 + K)
        M = int(alpha * N)
        F = r.standard_normal((M, N)) / np.sqrt(N)
        xs = [np.sqrt(v) * r.standard_normal(N) for v in pvars]
        y = sum(np.maximum(0, F @ x) for x in xs) + np.sqrt(nv) * r.standard_normal(M)
        experts = None
        for k in range(K):
            e = (GaussianPrior(size=N, var=pvars[k]) @ V(id=f"x_{k}") @ LinearChannel(F, name="F") @ V(id=f"z_{k}")
                 @ ReluChannel() @ V(id=f"a_{k}"))
            experts = e if experts is None else experts + e
        model = (experts @ SumChannel(n_prev=K) @ V(id="n") @ GaussianLikelihood(y=y, var=nv)).to_model()
        ep = ExpectationPropagation(model)
        iters = []

        def cb(algo, i, max_iter):
            rec = {}
            for source, target, data in algo.message_dag.edges(data=True):
                key = f"{nb_name(algo, source)}->{nb_name(algo, target)}"
                rec[key + ":a"] = float(data["a"])
                rec[key + ":b"] = np.array(data["b"], dtype=float)
            for vid, dd in algo.get_variables_data().items():
                rec[f"r:{vid}"] = np.array(dd["r"], dtype=float)
                rec[f"v:{vid}"] = float(dd["v"])
            iters.append(rec)
            return False
        err = None
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=12, callback=cb, damping=damping)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        sm = SumChannel(n_prev=K)
        azs, bzs = list(r.uniform(0.5, 2, K)), [r.standard_normal(5) for _ in range(K)]
        bxs = r.standard_normal(5)
        msgs = dict(az=azs, bz=bzs, ax=0.7, bx=bxs, fwd=sm.compute_forward_message(azs, bzs, 0.7, bxs),
                    bwd=sm.compute_backward_message(azs, bzs, 0.7, bxs))
        com.append(dict(K=K, N=N, M=M, pvars=pvars, noise_var=nv, damping=damping, xs=xs, y=y, error=err,   # F: first draw of RandomState(17 + K)
                        msgs=msgs, traj=pack_traj(iters),
                        mse=[float(np.mean((iters[-1][f"r:x_{k}"] - xs[k])**2)) for k in range(K)]))
        print("committee ep", K, N, damping, err, "mse", com[-1]["mse"])
    # state evolution of the soft committee (K = 2, SE is N-free: Marchenko-Pastur linear layers)
    comse = []
    for K, alpha, pvars, nv, damping in [(2, 2.0, [1.0, 0.5], 1e-2, None), (3, 1.5, [1.0, 0.6, 0.3], 5e-2, 0.2)]:
        experts = None
        for k in range(K):
            e = (GaussianPrior(size=None, var=pvars[k]) @ V(id=f"x_{k}") @ MarchenkoPasturChannel(alpha=alpha)
                 @ V(id=f"z_{k}") @ ReluChannel() @ V(id=f"a_{k}"))
            experts = e if experts is None else experts + e
        model = (experts @ SumChannel(n_prev=K) @ V(id="n") @ GaussianLikelihood(y=None, var=nv)).to_model()
        se = StateEvolution(model)
        its = []

        def cbse(algo, i, max_iter):
            rec = {}
            for source, target, data in algo.message_dag.edges(data=True):
                rec[f"{nb_name(algo, source)}->{nb_name(algo, target)}:a"] = float(data["a"])
            for vid, dd in algo.get_variables_data().items():
                rec[f"v:{vid}"] = float(dd["v"])
            its.append(rec)
            return False
        t0 = time.time()
        se.iterate(max_iter=6, callback=cbse, damping=damping)
        comse.append(dict(K=K, alpha=alpha, pvars=pvars, noise_var=nv, damping=damping, traj=pack_traj(its),
                          tau=model.get_second_moments()))
        print("committee se", K, damping, {k: v for k, v in its[-1].items() if k.startswith("v:")}, f"{time.time()-t0:.0f}s")
    # sign committee (committee_model.py:44-54): binary priors, sgn experts, majority vote
    from tramp.priors import BinaryPrior
    binp = []
    for p_pos in (0.5, 0.8):
        pr = BinaryPrior(size=9, p_pos=p_pos)
        bx = rs.standard_normal(9) * 2
        r_, v_ = pr.compute_forward_posterior(1.3, bx)
        a_, b_ = pr.compute_forward_message(1.3, bx)
        binp.append(dict(p_pos=p_pos, bx=bx, r=r_, v=v_, a_new=a_, b_new=b_))
    binse = [dict(p_pos=p, ax=a, v=BinaryPrior(size=None, p_pos=p).compute_forward_error(a))
             for p in (0.5, 0.8, 0.1) for a in (0.05, 0.7, 3.0, 25.0)]
    K, N, alpha, p_pos, damping = 3, 16, 4.0, [0.5, 0.6, 0.4], 0.5
    r = np.random.RandomState(29)
    M = int(alpha * N)
    F = r.standard_normal((M, N)) / np.sqrt(N)
    xs = [np.where(r.uniform(size=N) < p, 1.0, -1.0) for p in p_pos]
    y = np.sign(sum(np.sign(F @ x) for x in xs))
    experts = None
    for k in range(K):
        e = (BinaryPrior(size=N, p_pos=p_pos[k]) @ V(id=f"x_{k}") @ LinearChannel(F, name="F") @ V(id=f"z_{k}")
             @ SgnChannel() @ V(id=f"a_{k}"))
        experts = e if experts is None else experts + e
    model = (experts @ SumChannel(n_prev=K) @ V(id="a") @ SgnLikelihood(y=y)).to_model()
    ep = ExpectationPropagation(model)
    iters = []

    def cb2(algo, i, max_iter):
        rec = {}
        for source, target, data in algo.message_dag.edges(data=True):
            key = f"{nb_name(algo, source)}->{nb_name(algo, target)}"
            rec[key + ":a"] = float(data["a"])
            rec[key + ":b"] = np.array(data["b"], dtype=float)
        for vid, dd in algo.get_variables_data().items():
            rec[f"r:{vid}"] = np.array(dd["r"], dtype=float)
            rec[f"v:{vid}"] = float(dd["v"])
        iters.append(rec)
        return False
    err = None
    try:
        with np.errstate(all="ignore"):
            ep.iterate(max_iter=10, callback=cb2, damping=damping)
    except Exception as e:    # noqa: BLE001
        err = repr(e)
    sgncom = dict(K=K, N=N, M=M, p_pos=p_pos, damping=damping, xs=xs, y=y, error=err, traj=pack_traj(iters))   # F: first draw of RandomState(29)
    print("sgn committee ep", err, "iterations", len(iters), "overlaps", [float(np.mean(np.sign(iters[-1][f"r:x_{k}"]) == xs[k])) for k in range(K)])
    save("tree", data=np.array(grads, dtype=object), unit=np.array(unit, dtype=object),
         ep=np.array(trajs, dtype=object), reg=np.array(regs, dtype=object), gch=np.array(gch, dtype=object),
         se=np.array(ses, dtype=object), l21=np.array(l21, dtype=object), tv=np.array(tvs, dtype=object),
         vae=np.array(vae, dtype=object), com=np.array(com, dtype=object), binp=np.array(binp, dtype=object),
         sgncom=np.array([sgncom], dtype=object), comse=np.array(comse, dtype=object),
         binse=np.array(binse, dtype=object))


# ---------------------------------------------------------------- SE
def se_goldens(skip_slow):
    out = {}
    # 1-D quadrature rows (fast)
    gb = []
    for rho in (0.05, 0.1, 0.5):
        for (mean, var) in ((0., 1.), (0.3, 2.0)):
            p = GaussBernoulliPrior(size=None, rho=rho, mean=mean, var=var)
            for ax in (1e-3, 0.05, 0.5, 1.3, 10., 200.):
                gb.append(dict(rho=rho, mean=mean, var=var, ax=ax,
                               v=p.compute_forward_error(ax),
                               a_new=p.compute_forward_state_evolution(ax)))
    out["gb_prior"] = gb
    sg = []
    lk = SgnLikelihood(y=None)
    for (az, tau_z) in ((1.3, 1.0), (1.01, 1.0), (3.0, 0.5), (25., 2.0), (0.6, 2.0)):
        sg.append(dict(az=az, tau_z=tau_z, v=lk.compute_backward_error(az, tau_z),
                       a_new=lk.compute_backward_state_evolution(az, tau_z)))
    out["sgn_lik"] = sg
    rl = []
    lk = ReluLikelihood(y=None)
    for (az, tau_z) in ((1.3, 1.0), (3.0, 0.5), (25., 2.0)):
        rl.append(dict(az=az, tau_z=tau_z, v=lk.compute_backward_error(az, tau_z)))
    out["relu_lik"] = rl
    mp = []
    for alpha in (0.5, 0.7, 2.0):
        ch = MarchenkoPasturChannel(alpha=alpha)
        for (az, ax) in ((1.3, 0.7), (2.0, 0.), (1e-11, 5.0), (0., 5.0), (100., 100.), (0.3, 40.)):
            mp.append(dict(alpha=alpha, az=az, ax=ax,
                           vz=ch.compute_backward_error(az, ax, 1.0),
                           vx=ch.compute_forward_error(az, ax, 1.0),
                           tau_x=ch.second_moment(1.7),
                           mean_spectrum=ch.ensemble.mean_spectrum))
    out["mp"] = mp
    # SE trajectory: sparse GLM (GB prior + MP + Gaussian likelihood); fast
    traj = []
    for (alpha, rho, nv) in ((0.5, 0.05, 1e-2), (0.5, 0.25, 1e-2), (0.3, 0.25, 1e-8)):
        model = (GaussBernoulliPrior(size=None, rho=rho) @ V(id="x")
                 @ MarchenkoPasturChannel(alpha=alpha) @ V(id="z")
                 @ GaussianLikelihood(y=None, var=nv)).to_model()
        se = StateEvolution(model)
        rec = Recorder(EarlyStopping())
        se.iterate(max_iter=200, callback=rec)
        traj.append(dict(kind="glm_gauss", alpha=alpha, rho=rho, noise_var=nv, a0=0.,
                         n_iter=se.n_iter, traj=pack_traj(rec.iters)))
        print("se glm", alpha, rho, nv, se.n_iter)
    for (alpha, rho) in ((0.7, 0.25), (2.0, 0.5)):
        model = (GaussBernoulliPrior(size=None, rho=rho) @ V(id="x")
                 @ MarchenkoPasturChannel(alpha=alpha) @ V(id="z")
                 @ SgnLikelihood(y=None)).to_model()
        se = StateEvolution(model)
        rec = Recorder(EarlyStopping())
        a0 = 10.
        se.iterate(max_iter=200, callback=rec, initializer=ConstantInit(a=a0))
        traj.append(dict(kind="glm_sgn", alpha=alpha, rho=rho, a0=a0,
                         n_iter=se.n_iter, traj=pack_traj(rec.iters)))
        print("se sgn", alpha, rho, se.n_iter)
    out["traj"] = traj
    if not skip_slow:
        t0 = time.time()
        # 2-D quadrature rows (seconds each)
        ab = []
        lk = AbsLikelihood(y=None)
        for (az, tau_z) in ((1.3, 1.0), (3.0, 0.5)):
            ab.append(dict(az=az, tau_z=tau_z, v=lk.compute_backward_error(az, tau_z)))
        out["abs_lik"] = ab
        pw = []
        chans = dict(sgn=SgnChannel, abs=AbsChannel, relu=ReluChannel)
        recs = [(2.0, 1.5, 1.0), (2.1, 2.0, 2.0), (2.0, 1.5, 1.5), (1.3, 0., 1.0), (0.8, 2.5, 1.0),
                (1.0, 0.6, 1.0)]
        for kind, cls in chans.items():
            ch = cls()
            for (az, ax, tau_z) in recs:
                with np.errstate(all="ignore"):
                    vf = ch.compute_forward_error(az, ax, tau_z)
                    vb = ch.compute_backward_error(az, ax, tau_z)
                pw.append(dict(kind=kind, az=az, ax=ax, tau_z=tau_z, v_fwd=vf, v_bwd=vb,
                               tau_x=ch.second_moment(tau_z)))
                print("se pwl", kind, az, ax, tau_z, vf, vb, f"{time.time()-t0:.0f}s")
        out["pwl"] = pw
        pw2 = []
        for kind, param, ch in [("l-relu", 0.1, LeakyReluChannel(slope=0.1)),
                                ("a-abs", 0.3, AsymmetricAbsChannel(shift=0.3))]:
            for (az, ax, tau_z) in recs[:4]:
                with np.errstate(all="ignore"):
                    vf = ch.compute_forward_error(az, ax, tau_z)
                    vb = ch.compute_backward_error(az, ax, tau_z)
                pw2.append(dict(kind=kind, param=param, az=az, ax=ax, tau_z=tau_z, v_fwd=vf, v_bwd=vb,
                                tau_x=ch.second_moment(tau_z)))
                print("se pwl2", kind, az, ax, tau_z, vf, vb, f"{time.time()-t0:.0f}s")
        out["pwl2"] = pw2
        pw3 = []
        for kind, param, ch in [("h-tanh", None, HardTanhChannel()), ("door", 0.8, SymmetricDoorChannel(width=0.8))]:
            for (az, ax, tau_z) in recs[:3]:
                with np.errstate(all="ignore"):
                    vf = ch.compute_forward_error(az, ax, tau_z)
                    vb = ch.compute_backward_error(az, ax, tau_z)
                pw3.append(dict(kind=kind, param=param, az=az, ax=ax, tau_z=tau_z, v_fwd=vf, v_bwd=vb,
                                tau_x=ch.second_moment(tau_z)))
                print("se pwl3", kind, az, ax, tau_z, vf, vb, f"{time.time()-t0:.0f}s", flush=True)
        out["pwl3"] = pw3
        # two-layer relu SE, a few iterations (13 s / iteration in the reference)
        model = (GaussianPrior(size=None) @ V(id="x") @ MarchenkoPasturChannel(alpha=2.0)
                 @ V(id="z1") @ ReluChannel() @ V(id="a1")
                 @ MarchenkoPasturChannel(alpha=0.5) @ V(id="z2")
                 @ GaussianLikelihood(y=None, var=1e-2)).to_model()
        se = StateEvolution(model)
        rec = Recorder(EarlyStopping())
        with np.errstate(all="ignore"):
            se.iterate(max_iter=6, callback=rec)
        out["traj_two_layer"] = dict(alpha1=2.0, alpha2=0.5, noise_var=1e-2, a0=0.,
                                     n_iter=se.n_iter, traj=pack_traj(rec.iters),
                                     tau=model.get_second_moments())
        print("se two-layer", se.n_iter, f"{time.time()-t0:.0f}s")
    save("se", data=np.array(out, dtype=object))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-slow", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    only = set(args.only.split(",")) if args.only else None
    if not only or "factors" in only:
        factor_goldens()
    if not only or "linear" in only:
        linear_goldens()
    if not only or "ep" in only:
        ep_goldens()
    if not only or "conv" in only:
        conv_goldens()
    if not only or "pwl3lik" in only:
        pwl3_lik_goldens()
    if not only or "unitary" in only:
        unitary_goldens()
    if not only or "logz" in only:
        logz_goldens()
    if not only or "tree" in only:
        tree_goldens()
    if not only or "se" in only:
        se_goldens(args.skip_slow)
