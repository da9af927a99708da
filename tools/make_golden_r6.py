#!/usr/bin/env python3
"""Golden vectors for the factors added in round 6, computed by IMPORTING the reference:

* ExponentialPrior, PositivePrior, GaussianMixturePrior (tramp/priors/{exponential,positive,gaussian_mixture}_prior.py):
  posteriors, messages, log-partitions, diagonal beliefs, the mixture's SE error, EP trajectories behind a dense
  LinearChannel (every message of every iteration; W is re-drawn from the recorded seed by the tests:
  RandomState(1000 seed + N).randn(M, N) / sqrt(N) FIRST, as here)            -> tests/golden/priors2.npz
* ModulusLikelihood (tramp/likelihoods/modulus_likelihood.py) and ComplexLinearChannel
  (tramp/channels/linear/complex_linear_channel.py): function level, and the phase-retrieval EP trajectories
  GaussBernoulliPrior(size=(2, N)) -> ComplexLinearChannel -> ModulusLikelihood     -> tests/golden/complex.npz

Runs ONLY in the build container (needs /root/reference); the outputs are data (inputs and the reference's outputs).

    python tools/make_golden_r6.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the networkx-1.x shim and puts the reference on sys.path)

from tramp.priors import ExponentialPrior, PositivePrior, GaussianMixturePrior, GaussBernoulliPrior  # noqa: E402
from tramp.likelihoods import GaussianLikelihood, ModulusLikelihood  # noqa: E402
from tramp.channels import LinearChannel, ComplexLinearChannel  # noqa: E402
from tramp.variables import SISOVariable as V  # noqa: E402
from tramp.algos import ExpectationPropagation, EarlyStoppingEP, CustomInit  # noqa: E402

MIXES = [dict(probs=[0.5, 0.5], means=[-1, 1], vars=[1, 1]),
         dict(probs=[0.2, 0.5, 0.3], means=[-2.0, 0.0, 1.5], vars=[0.3, 0.01, 2.0]),
         dict(probs=[0.9, 0.1], means=[0.0, 0.0], vars=[1e-4, 1.0])]


def edge_b(rs, n, scale):
    b = scale * rs.standard_normal(n)
    b[:8] = [0.0, -0.0, 1e-300, -1e-12, 40.0, -40.0, 300.0, -300.0]
    return b


def prior_cases():
    rs = np.random.RandomState(20261004)
    n = 261
    out = []
    specs = [("exponential", dict(mean=1), ExponentialPrior), ("exponential", dict(mean=0.5), ExponentialPrior),
             ("exponential", dict(mean=3.0), ExponentialPrior), ("positive", {}, PositivePrior)]
    specs += [("mixture", m, GaussianMixturePrior) for m in MIXES]
    for name, kw, cls in specs:
        pr = cls(size=n, **kw)
        for ax, sb in [(1.3, 1.5), (0.05, 0.3), (40.0, 25.0), (1e-3, 0.05), (3e3, 80.0)]:
            bx = edge_b(rs, n, sb)
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                r, v = pr.compute_forward_posterior(ax, bx)
                a_new, b_new = pr.compute_forward_message(ax, bx)
                A = pr.compute_log_partition(ax, bx)
            out.append(dict(name=name, kwargs=kw, ax=ax, bx=bx, r=np.asarray(r), v=float(v), a_new=float(a_new),
                            b_new=np.asarray(b_new), A=float(A), second_moment=float(pr.second_moment())))
        # diagonal beliefs (isotropic=False): the mode of the reference's own prior tests (tests/test_priors.py:17-51)
        prd = cls(size=n, isotropic=False, **kw)
        axd = np.exp(rs.uniform(np.log(0.05), np.log(30.0), n))
        bxd = axd * rs.standard_normal(n) * 1.5
        with np.errstate(all="ignore"):
            r, v = prd.compute_forward_posterior(axd, bxd)
            a_new, b_new = prd.compute_forward_message(axd, bxd)
        out.append(dict(name=name, kwargs=kw, diag=True, ax=axd, bx=bxd, r=np.asarray(r), v=np.asarray(v),
                        a_new=np.asarray(a_new), b_new=np.asarray(b_new)))
        print(name, kw, "ok")
    return out


def se_cases():
    out = []
    for m in MIXES:
        pr = GaussianMixturePrior(size=None, **m)
        for ax in (1e-2, 0.3, 1.3, 7.0, 60.0):
            out.append(dict(name="mixture", kwargs=m, ax=ax, v=float(pr.compute_forward_error(ax)),
                            a_new=float(pr.compute_forward_state_evolution(ax))))
            print("se mixture", m["probs"], ax, out[-1]["v"])
    return out


def sample_prior(rs, name, kw, N):
    if name == "exponential":
        return rs.exponential(scale=kw.get("mean", 1), size=N)
    if name == "positive":
        return np.abs(rs.standard_normal(N))
    k = rs.choice(len(kw["probs"]), size=N, p=kw["probs"])
    return np.asarray(kw["means"], float)[k] + np.sqrt(np.asarray(kw["vars"], float))[k] * rs.standard_normal(N)


def ep_cases():
    out = []
    classes = dict(exponential=ExponentialPrior, positive=PositivePrior, mixture=GaussianMixturePrior)
    runs = [("exponential", dict(mean=1), 0.7, None, 1e-2), ("exponential", dict(mean=0.5), 0.5, 0.3, 1e-2),
            ("positive", {}, 0.7, None, 1e-2), ("positive", {}, 0.5, 0.2, 5e-2),
            ("mixture", MIXES[0], 0.8, None, 1e-2), ("mixture", MIXES[1], 0.6, 0.1, 1e-2),
            ("mixture", MIXES[2], 0.5, None, 1e-3)]
    for name, kw, alpha, damping, nv in runs:
        for N, seed in ((64, 1), (200, 2)):
            rs = np.random.RandomState(1000 * seed + N)
            M = int(alpha * N)
            W = rs.randn(M, N) / np.sqrt(N)
            x = sample_prior(rs, name, kw, N)
            y = W @ x + np.sqrt(nv) * rs.standard_normal(M)
            # the exponential prior's first message from a = 0 is NaN in the reference: start from a small precision
            init = CustomInit(a=1e-3 if name == "exponential" else 0, b=0)
            model = (classes[name](size=N, **kw) @ V(id="x") @ LinearChannel(W) @ V(id="z")
                     @ GaussianLikelihood(y=y, var=nv)).to_model()
            ep = ExpectationPropagation(model)
            rec = mg.Recorder(EarlyStoppingEP())
            err = None
            try:
                with warnings.catch_warnings(), np.errstate(all="ignore"):
                    warnings.simplefilter("ignore")
                    ep.iterate(max_iter=30, callback=rec, damping=damping, initializer=init)
            except Exception as e:   # noqa: BLE001
                err = repr(e)
            out.append(dict(name=name, kwargs=kw, N=N, M=M, alpha=alpha, seed=seed, damping=damping, noise_var=nv,
                            a_init=init.a, W_sum=float(W.sum()), x=x, y=y, n_iter=ep.n_iter, error=err, n_recorded=len(rec.iters),
                            traj=mg.pack_traj(rec.iters[:12])))
            print("ep", name, N, damping, "n_iter", ep.n_iter, err)
    return out


def complex_cases():
    rs = np.random.RandomState(20261005)
    n = 203
    mod = []
    for az, sb in [(1.3, 1.5), (0.2, 0.4), (25.0, 30.0), (2.0, 3e4)]:
        bz = sb * rs.standard_normal((2, n))
        y = np.abs(rs.standard_normal(n) + 1j * rs.standard_normal(n))
        bz[:, 0] = 0.0                      # normalize(0) = 0
        y[1] = 0.0
        bz[:, 2] = [3e5, -4e5]              # |bz| y beyond the 1e9 cut-off with y[2] below
        y[2] = 5e3
        lk = ModulusLikelihood(y=y)
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            r, v = lk.compute_backward_posterior(az, bz, y)
            a_new, b_new = lk.compute_backward_message(az, bz)
            A = lk.compute_log_partition(az, bz, y)
        mod.append(dict(az=az, bz=bz, y=y, r=np.asarray(r), v=float(v), a_new=float(a_new), b_new=np.asarray(b_new),
                        A=float(A)))
    lin = []
    for (M, N) in [(24, 16), (16, 24), (20, 20)]:
        W = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))) / np.sqrt(2 * N)
        if (M, N) == (20, 20):
            W[:, -1] = W[:, 0]             # rank deficient
        ch = ComplexLinearChannel(W)
        for az, ax in [(1.3, 0.7), (0.05, 4.0), (2.0, 0.0)]:
            bz = 1.5 * rs.standard_normal((2, N))
            bx = 1.5 * rs.standard_normal((2, M)) * (ax > 0)
            with np.errstate(all="ignore"):
                rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
                rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
                ab = ch.compute_backward_message(az, bz, ax, bx)
                af = ch.compute_forward_message(az, bz, ax, bx)
                A = ch.compute_log_partition(az, bz, ax, bx)
            lin.append(dict(M=M, N=N, W=W, az=az, ax=ax, bz=bz, bx=bx, rz=rz, vz=float(vz), rx=rx, vx=float(vx),
                            a_bwd=float(ab[0]), b_bwd=np.asarray(ab[1]), a_fwd=float(af[0]), b_fwd=np.asarray(af[1]),
                            A=float(A), rank=int(ch.rank), n_eff=float(ch.compute_n_eff(az, ax))))
    ep = []
    for N, alpha, rho, damping, seed in [(48, 2.0, 0.6, 0.3, 1), (96, 3.0, 1.0 - 1e-9, None, 2), (64, 2.5, 0.5, 0.5, 3)]:
        rs2 = np.random.RandomState(77 * seed + N)
        M = int(alpha * N)
        W = (rs2.standard_normal((M, N)) + 1j * rs2.standard_normal((M, N))) / np.sqrt(2 * N)
        x = rs2.standard_normal((2, N)) * rs2.binomial(1, min(rho, 1.0), (2, N))
        y = np.abs(W @ (x[0] + 1j * x[1]))
        b0 = 0.5 * rs2.standard_normal((2, N))         # break the phase symmetry: a non-zero first message to the prior
        init = CustomInit(a_init=[("x", "bwd", 0.1)], b_init=[("x", "bwd", b0)], a=0, b=0)
        model = (GaussBernoulliPrior(size=(2, N), rho=rho) @ V(id="x") @ ComplexLinearChannel(W) @ V(id="z")
                 @ ModulusLikelihood(y=y)).to_model()
        algo = ExpectationPropagation(model)
        rec = mg.Recorder(EarlyStoppingEP())
        err = None
        try:
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                algo.iterate(max_iter=25, callback=rec, damping=damping, initializer=init)
        except Exception as e:   # noqa: BLE001
            err = repr(e)
        ep.append(dict(N=N, M=M, alpha=alpha, rho=rho, damping=damping, seed=seed, W_sum=complex(W.sum()), x=x, y=y, b0=b0, a0=0.1,
                       n_iter=algo.n_iter, error=err, n_recorded=len(rec.iters), traj=mg.pack_traj(rec.iters[:12])))
        print("phase retrieval", N, alpha, damping, "n_iter", algo.n_iter, err)
    return mod, lin, ep


def modulus_se_cases():
    "ModulusLikelihood.compute_backward_error on a grid, and the SE trajectory of phase retrieval (alpha = 2.5, rho = 0.5)"
    from tramp.models import glm_state_evolution
    from tramp.algos import StateEvolution, EarlyStopping, ConstantInit
    lk = ModulusLikelihood(y=None)
    grid = []
    for az, tau in [(1.3, 1.0), (2.0, 1.0), (0.5, 2.0), (1.0, 1.0), (5.0, 1.0), (1.05, 1.0), (40.0, 1.0), (0.9, 3.0),
                    (300.0, 0.5), (1.0 + 1e-6, 1.0)]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            grid.append(dict(az=az, tau_z=tau, v=float(lk.compute_backward_error(az, tau)),
                             a_new=float(lk.compute_backward_state_evolution(az, tau))))
        print("se modulus", az, tau, grid[-1]["v"])
    trajs = []
    for alpha, rho, a0 in [(2.5, 0.5, 0.1), (1.6, 0.3, 1.0)]:
        model = glm_state_evolution(alpha=alpha, prior_type="gauss_bernoulli", output_type="modulus", prior_rho=rho)
        se = StateEvolution(model)
        rec = mg.Recorder(EarlyStopping())
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            se.iterate(max_iter=60, callback=rec, initializer=ConstantInit(a=a0))
        trajs.append(dict(alpha=alpha, rho=rho, a0=a0, n_iter=se.n_iter, tau=model.get_second_moments(),
                          traj=mg.pack_traj(rec.iters)))
        print("se phase retrieval", alpha, rho, "n_iter", se.n_iter, rec.iters[-1])
    return grid, trajs


def main():
    mg.save("priors2", cases=np.array(prior_cases(), dtype=object), se=np.array(se_cases(), dtype=object),
            ep=np.array(ep_cases(), dtype=object))
    mod, lin, ep = complex_cases()
    grid, trajs = modulus_se_cases()
    mg.save("complex", modulus=np.array(mod, dtype=object), linear=np.array(lin, dtype=object),
            ep=np.array(ep, dtype=object), se=np.array(grid, dtype=object), se_traj=np.array(trajs, dtype=object))


if __name__ == "__main__":
    main()
