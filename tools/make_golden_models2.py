#!/usr/bin/env python3
"""Golden vectors for the model builders and ensembles completed late in round 6, computed by IMPORTING the reference:

* the three remaining builders of tramp/models/total_variation_model.py - sparse_gradient_classification (:74-78),
  tv_regression (:81-85), tv_classification (:88-92) - built here BY HAND from the reference's factor classes, because
  the module itself cannot run (np.product is gone from NumPy 2, and tv_block passes scale= to MAP_L21NormPrior, whose
  argument is gamma): every message and posterior of every EP iteration;
* DuplicateChannel (tramp/channels/shape/duplicate_channel.py): posterior and messages, and EP on a signal observed twice;
* MarchenkoPasturEnsemble.eta_transform / shannon_transform / measure (tramp/ensembles/marchenko_pastur_ensemble.py:31-53).

Runs ONLY in the build container (needs /root/reference); the output is data.   -> tests/golden/models2.npz

    python tools/make_golden_models2.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the networkx-1.x shim and puts the reference on sys.path)

from tramp.priors import GaussianPrior, GaussBernoulliPrior, MAP_L21NormPrior  # noqa: E402
from tramp.likelihoods import GaussianLikelihood, SgnLikelihood  # noqa: E402
from tramp.channels import LinearChannel, GaussianChannel, GradientChannel, ReshapeChannel  # noqa: E402
from tramp.variables import SISOVariable as V, SIMOVariable, MILeafVariable  # noqa: E402
from tramp.algos import ExpectationPropagation, EarlyStoppingEP, ConstantInit  # noqa: E402
from tramp.ensembles import MarchenkoPasturEnsemble  # noqa: E402


def signal(r, shape, rho):
    if len(shape) == 1:
        x = np.cumsum(r.standard_normal(shape) * r.binomial(n=1, size=shape, p=rho))
        return x - x.mean()
    return np.kron(r.standard_normal((2, 2)), np.ones((shape[0] // 2, shape[1] // 2)))


def model_cases():
    out = []
    specs = [("sparse_gradient_classification", (48,), 60, dict(grad_rho=0.08), 0.05, 0.2),
             ("sparse_gradient_classification", (6, 8), 72, dict(grad_rho=0.2), 0.1, 0.3),
             ("tv_regression", (48,), 30, dict(grad_scale=0.5), 1e-2, 0.3),
             ("tv_regression", (6, 8), 30, dict(grad_scale=0.4), 1e-2, 0.3),
             ("tv_classification", (48,), 60, dict(grad_scale=0.5), 0.05, 0.3),
             ("tv_classification", (6, 8), 72, dict(grad_scale=0.4), 0.1, 0.3)]
    for name, shape, M, kw, nv, damping in specs:
        r = np.random.RandomState(17 + len(out))
        d, N = len(shape), int(np.prod(shape))
        x = signal(r, shape, 0.08)
        A = r.standard_normal((M, N)) / np.sqrt(N)
        z = A @ x.ravel()
        classify = name.endswith("classification")
        y = np.sign(z + np.sqrt(nv) * r.standard_normal(M)) if classify else z + np.sqrt(nv) * r.standard_normal(M)
        if name.startswith("tv"):
            grad_prior = MAP_L21NormPrior(size=(d,) + shape, gamma=kw["grad_scale"], axis=0)
        else:
            grad_prior = GaussBernoulliPrior(size=(d,) + shape, rho=kw["grad_rho"])
        block = (GaussianPrior(size=shape, var=1.0) @ SIMOVariable(id="x", n_next=2) @ (
            (GradientChannel(shape=shape) + grad_prior) @ MILeafVariable(id="x'", n_prev=2))
        ) @ ReshapeChannel(prev_shape=shape, next_shape=N)
        if classify:
            tail = LinearChannel(A, name="A") @ V(id="z") @ GaussianChannel(var=nv) @ V(id="a") @ SgnLikelihood(y)
        else:
            tail = LinearChannel(A, name="A") @ V(id="z") @ GaussianLikelihood(y, var=nv)
        model = (block @ V(id="r") @ tail).to_model()
        ep = ExpectationPropagation(model)
        rec = mg.TreeRecorder(EarlyStoppingEP())
        err = None
        # b = 0 makes every group norm of the L21 prior 0 and the reference's variance 0 * nan: start from a constant b
        init = ConstantInit(a=0.1, b=1.0)
        try:
            with np.errstate(all="ignore"):
                ep.iterate(max_iter=30, callback=rec, damping=damping, initializer=init)
        except Exception as e:    # noqa: BLE001
            err = repr(e)
        out.append(dict(name=name, shape=shape, M=M, kwargs=kw, noise_var=nv, damping=damping, a0=0.1, b0=1.0, x=x, y=y, A=A,
                        n_iter=ep.n_iter, error=err, n_recorded=len(rec.iters), traj=mg.pack_traj(rec.iters[:10]),
                        final_r=rec.iters[-1]["r:x"], mse=float(np.mean((rec.iters[-1]["r:x"] - x)**2))))
        print(name, shape, "n_iter", ep.n_iter, err, "mse", out[-1]["mse"], "var x", x.var())
    return out


def duplicate_cases():
    """DuplicateChannel (tramp/channels/shape/duplicate_channel.py): function level, and EP on a signal seen twice -
    directly through Gaussian noise and through a dense matrix (every message of every iteration)."""
    from tramp.channels import DuplicateChannel
    rs = np.random.RandomState(23)
    fn = []
    for K, n in [(2, 33), (3, 17)]:
        ch = DuplicateChannel(n_next=K)
        az, bz = 1.3, rs.standard_normal(n) * 2
        ax = [float(a) for a in rs.uniform(0.2, 3.0, K)]
        bx = [rs.standard_normal(n) for _ in range(K)]
        rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
        ab, bb = ch.compute_backward_message(az, bz, ax, bx)
        # the forward messages as Factor.forward_message forms them (base.py:329-346): compute_ab_new per output
        fwd = [ch.compute_ab_new(rz, vz, a, b) for a, b in zip(ax, bx)]
        fn.append(dict(K=K, az=az, bz=bz, ax=ax, bx=bx, rz=rz, vz=float(vz), a_bwd=float(ab), b_bwd=bb,
                       a_fwd=[float(f[0]) for f in fwd], b_fwd=[f[1] for f in fwd],
                       v_bwd_err=float(ch.compute_backward_error(az, ax, 1.0))))
    eps = []
    for N, M, rho, nv1, nv2, damping in [(40, 25, 0.2, 0.5, 0.05, None), (64, 30, 0.1, 1.0, 0.02, 0.3)]:
        r = np.random.RandomState(41 + N)
        x = r.standard_normal(N) * r.binomial(1, rho, N)
        A = r.standard_normal((M, N)) / np.sqrt(N)
        y1 = x + np.sqrt(nv1) * r.standard_normal(N)
        y2 = A @ x + np.sqrt(nv2) * r.standard_normal(M)
        model = (GaussBernoulliPrior(size=N, rho=rho) @ V(id="z") @ DuplicateChannel(n_next=2) @ (
            V(id="x1") @ GaussianLikelihood(y1, var=nv1)
            + V(id="x2") @ LinearChannel(A, name="A") @ V(id="w") @ GaussianLikelihood(y2, var=nv2))).to_model()
        ep = ExpectationPropagation(model)
        rec = mg.TreeRecorder(EarlyStoppingEP())
        with np.errstate(all="ignore"):
            ep.iterate(max_iter=30, callback=rec, damping=damping)
        eps.append(dict(N=N, M=M, rho=rho, nv1=nv1, nv2=nv2, damping=damping, x=x, A=A, y1=y1, y2=y2, n_iter=ep.n_iter,
                        traj=mg.pack_traj(rec.iters[:10]), final_r=rec.iters[-1]["r:z"],
                        mse=float(np.mean((rec.iters[-1]["r:z"] - x)**2))))
        print("duplicate ep", N, damping, "n_iter", ep.n_iter, "mse", eps[-1]["mse"], "vs y1", float(np.mean((y1 - x)**2)))
    return fn, eps


def mp_cases():
    out = []
    for alpha in (0.3, 0.5, 1.0, 2.5):
        ens = MarchenkoPasturEnsemble(alpha=alpha)
        for gamma in (1e-3, 0.1, 1.0, 7.0, 1e3):
            out.append(dict(alpha=alpha, gamma=gamma, eta=float(ens.eta_transform(gamma)), shannon=float(ens.shannon_transform(gamma)),
                            mean_spectrum=float(ens.mean_spectrum), z_min=float(ens.z_min), z_max=float(ens.z_max),
                            measure_sq=float(ens.measure(lambda z: z * z))))
    return out


if __name__ == "__main__":
    models = model_cases()
    mps = mp_cases()
    dup_fn, dup_ep = duplicate_cases()
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "models2.npz")
    np.savez_compressed(path, models=np.array(models, dtype=object), mp=np.array(mps, dtype=object),
                        dup_fn=np.array(dup_fn, dtype=object), dup_ep=np.array(dup_ep, dtype=object))
    print("wrote", path, os.path.getsize(path), "bytes")
