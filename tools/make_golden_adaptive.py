#!/usr/bin/env python3
"""Golden trajectories of EP with ``damping="adaptive"`` (tramp/algos/message_passing.py:129-185): sparse GLMs
(GaussBernoulli prior, dense LinearChannel, Gaussian / sign / abs likelihood) for a few iterations, ALL FOUR messages of
every variable (factor -> variable and the separately damped variable -> factor pass-through edges), posteriors and the
step beta the reference chose on every edge.

The step search compares log-partition differences with zero, so a decision can hinge on round-off when a candidate barely
changes the objective.  The script replays every decision and keeps only configurations in which each decision that
matters (the new message differs from the old one) has |dA| well away from zero for every candidate examined - such
trajectories are reproducible by any faithful implementation, whatever its summation order.

Runs ONLY in the build container (imports /root/reference); writes tests/golden/adaptive.npz.

    python tools/make_golden_adaptive.py
"""
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, HERE)
from make_golden import install_networkx1_shim  # noqa: E402,F401  (importing it installs the networkx 1.x shim)

from tramp.algos import ExpectationPropagation  # noqa: E402
from tramp.algos.message_passing import create_message  # noqa: E402
from tramp.base import Variable  # noqa: E402
from tramp.channels import LinearChannel  # noqa: E402
from tramp.likelihoods import AbsLikelihood, GaussianLikelihood, SgnLikelihood  # noqa: E402
from tramp.priors import GaussBernoulliPrior  # noqa: E402
from tramp.variables import SISOVariable as V  # noqa: E402

logging.disable(logging.CRITICAL)
MARGIN = 1e-6          # |dA| of every examined candidate of a decision that matters
SIGNIFICANT = 1e-7     # |new - old| (max over a and b) above which a decision matters


def replay_margin(algo, source, target, data):
    "the candidates compute_adaptive_damping examines for this edge: (significant?, smallest |dA| examined, beta)"
    if algo.n_iter == 0:
        return False, np.inf, 1.0
    variable = target if isinstance(target, Variable) else source
    old = algo.message_dag[source][target]
    var = {k: data[k] - old[k] for k in ("a", "b")}
    sig = max(abs(var["a"]), float(np.max(np.abs(var["b"]))))
    m_t = algo.message_dag.in_edges(target, data=True)
    m_e = [(source, target, old), (target, source, algo.message_dag[target][source])]
    A_old = algo.node_objective(target, m_t) - algo.node_objective(variable, m_e)
    worst = np.inf
    for n in range(10):
        beta = 1 / 2**n
        new = dict(data)
        for k in ("a", "b"):
            new[k] = old[k] + beta * var[k]
        dA = (algo.node_objective(target, create_message(m_t, source, target, new))
              - algo.node_objective(variable, create_message(m_e, source, target, new))) - A_old
        # inf - inf (a variable with a <= 0) compares False whatever the round-off; an exact zero is what identical
        # old and new messages give (a factor re-evaluated on unchanged inputs), again independent of round-off
        if np.isfinite(dA) and dA != 0:
            worst = min(worst, abs(dA))
        if dA >= 0:
            return sig > SIGNIFICANT, worst, beta
    return sig > SIGNIFICANT, worst, 0.0


def run_case(N, alpha, output, seed, rho, n_iter):
    rs = np.random.RandomState(seed)
    M = int(alpha * N)
    W = rs.randn(M, N) / np.sqrt(N)
    x = rs.randn(N) * (rs.rand(N) < rho)
    z = W @ x
    if output == "sgn":
        y, lik = np.sign(z), SgnLikelihood(y=np.sign(z))
    elif output == "abs":
        y, lik = np.abs(z), AbsLikelihood(y=np.abs(z))
    else:
        y = z + 0.1 * rs.randn(M)
        lik = GaussianLikelihood(y=y, var=0.01)
    model = (GaussBernoulliPrior(size=N, rho=rho) @ V(id="x") @ LinearChannel(W) @ V(id="z") @ lik).to_model()
    ep = ExpectationPropagation(model)
    margins, betas = [], []
    orig = ep.compute_adaptive_damping

    def wrapped(source, target, data):
        sig, worst, beta = replay_margin(ep, source, target, data)
        out = orig(source, target, data)
        if ep.n_iter > 0:
            assert out.get("beta", 1.0) == beta, (out.get("beta"), beta)
            if sig:
                margins.append(worst)
            betas.append((ep.n_iter, getattr(source, "id", None) or repr(source), getattr(target, "id", None) or repr(target),
                          beta, bool(sig)))
        return out

    ep.compute_adaptive_damping = wrapped
    traj = []

    def record(algo, i, max_iter):
        rec = {}
        for var in algo.variables:
            for s, t, d in algo.message_dag.in_edges(var, data=True):          # factor -> variable
                rec[f"{d['direction']}:f->{var.id}"] = (float(d["a"]), np.array(d["b"]))
            for s, t, d in algo.message_dag.out_edges(var, data=True):         # variable -> factor
                rec[f"{d['direction']}:{var.id}->f"] = (float(d["a"]), np.array(d["b"]))
            nd = algo.message_dag.node[var]
            rec[f"r:{var.id}"], rec[f"v:{var.id}"] = np.array(nd["r"]), float(nd["v"])
        traj.append(rec)
        return False

    ep.iterate(max_iter=n_iter, callback=record, damping="adaptive")
    return dict(N=N, alpha=alpha, output=output, seed=seed, rho=rho, n_iter=n_iter, W_sum=float(W.sum()), y=y, traj=traj,
                betas=betas), (min(margins) if margins else np.inf), sum(1 for b in betas if b[3] < 1 and b[4])


def main():
    cases = []
    for output, alpha in [("gaussian", 0.5), ("sgn", 1.5), ("abs", 1.5), ("gaussian", 1.5)]:
        for seed in range(1, 60):
            c, margin, n_damped = run_case(60, alpha, output, seed, 0.3, 8)
            if margin > MARGIN and n_damped >= 3:
                print(output, alpha, "seed", seed, "smallest |dA| among decisions that matter", margin,
                      "edges damped (beta < 1, message changing):", n_damped)
                cases.append(c)
                break
        else:
            raise SystemExit(f"no robust seed for {output} alpha={alpha}")
    np.savez_compressed(os.path.join(OUT, "adaptive.npz"), data=np.array(cases, dtype=object))
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
