"""Pin the oracle's round-6 restatements (Exponential / Positive / GaussianMixture priors, ModulusLikelihood,
ComplexLinearChannel) on golden vectors captured by importing the reference (tools/make_golden_r6.py)."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc
from tests.conftest import prior2_oracle, redraw_prior2_W, redraw_complex_W


def close(a, b, rtol=1e-12, atol=0.0, msg=""):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def test_prior2_function_level(golden_priors2):
    n_iso = n_diag = 0
    for c in golden_priors2["cases"]:
        post, _, logz = prior2_oracle(orc, c["name"], c["kwargs"])
        tag = str((c["name"], c["kwargs"], np.ndim(c["ax"]) and "diag" or c["ax"]))
        with np.errstate(all="ignore"):
            if c.get("diag"):
                r, v = post(c["ax"], c["bx"], isotropic=False)
                close(r, c["r"], 1e-13, 1e-300, tag)
                close(v, c["v"], 1e-12, 1e-300, tag)
                a_new, b_new = orc.ab_new(r, v, c["ax"], c["bx"])
                close(a_new, c["a_new"], 1e-10, 0, tag)
                close(b_new, c["b_new"], 1e-10, 1e-12 * np.abs(c["b_new"]).max(), tag)
                n_diag += 1
                continue
            r, v = post(c["ax"], c["bx"])
            a_new, b_new = orc.ab_new(r, v, c["ax"], c["bx"])
            A = logz(c["ax"], c["bx"])
        np.testing.assert_array_equal(np.isnan(r), np.isnan(c["r"]), err_msg=tag)
        close(r, c["r"], 1e-13, 1e-300, tag)
        close(v, c["v"], 1e-13, 0, tag)
        close(a_new, c["a_new"], 1e-11, 0, tag)
        close(b_new, c["b_new"], 1e-11, 1e-13 * np.nanmax(np.abs(c["b_new"])), tag)
        close(A, c["A"], 1e-12, 0, tag)
        n_iso += 1
    assert (n_iso, n_diag) == (35, 7)


def test_mixture_se_error(golden_priors2):
    for c in golden_priors2["se"]:
        kw = c["kwargs"]
        v = orc.se_gm_prior_error(c["ax"], kw["probs"], kw["means"], kw["vars"])
        close(v, c["v"], 1e-9)
        close(orc.a_new(v, c["ax"]), c["a_new"], 1e-8)
        vg = orc.se_gm_prior_error(c["ax"], kw["probs"], kw["means"], kw["vars"], method="gl")
        close(vg, c["v"], 2e-6)


def _check_traj(recs, t, ids, n_cap=12):
    n = min(t[f"v:{ids[0]}"].shape[0], n_cap)
    for it in range(n):
        for i, vid in enumerate(ids):
            tol = 1e-7 if it < 6 else 1e-5
            close(recs[it]["fwd"][i][0], t[f"fwd:{vid}:a"][it], tol)
            close(recs[it]["bwd"][i][0], t[f"bwd:{vid}:a"][it], tol)
            for key, got in (("fwd", recs[it]["fwd"][i][1]), ("bwd", recs[it]["bwd"][i][1])):
                ref = t[f"{key}:{vid}:b"][it]
                close(got, ref, tol, tol * np.abs(ref).max())
            ref = t[f"r:{vid}"][it]
            close(recs[it]["r"][i], ref, tol, tol * np.abs(ref).max())
            close(recs[it]["v"][i], t[f"v:{vid}"][it], tol)


def _recorder():
    recs = []

    def record(it, fwd, bwd, rs_, vs_):
        recs.append(dict(fwd=[(a, np.array(b, float)) for a, b in fwd], bwd=[(a, np.array(b, float)) for a, b in bwd],
                         r=[np.array(r) for r in rs_], v=list(vs_)))
    return recs, record


def test_prior2_ep_trajectories(golden_priors2):
    assert len(golden_priors2["ep"]) == 14
    for c in golden_priors2["ep"]:
        assert c["error"] is None
        W = redraw_prior2_W(c)
        _, factor, _ = prior2_oracle(orc, c["name"], c["kwargs"])
        fs = [factor(c["N"]), orc.LinearFactor(W), orc.GaussLikFactor(c["y"], c["noise_var"])]
        recs, record = _recorder()
        with np.errstate(all="ignore"):
            res = orc.ep_chain(fs, [c["N"], c["M"]], max_iter=30, damping=c["damping"], record=record,
                               init=lambda i, d: (c["a_init"], 0.0))
        assert res["n_iter"] == c["n_iter"], (c["name"], c["N"], c["damping"])
        _check_traj(recs, c["traj"], ["x", "z"])


def test_modulus_likelihood(golden_complex):
    for c in golden_complex["modulus"]:
        with np.errstate(all="ignore"):
            r, v = orc.modulus_lik_posterior(c["az"], c["bz"], c["y"])
            a_new, b_new = orc.modulus_lik_message(c["az"], c["bz"], c["y"])
            A = orc.modulus_lik_log_partition(c["az"], c["bz"], c["y"])
        close(r, c["r"], 1e-13, 1e-300)
        close(v, c["v"], 1e-13)
        close(a_new, c["a_new"], 1e-11)
        close(b_new, c["b_new"], 1e-11, 1e-13 * np.abs(c["b_new"]).max())
        np.testing.assert_allclose(A, c["A"], rtol=1e-12, equal_nan=True)


def test_complex_linear_channel(golden_complex):
    assert len(golden_complex["linear"]) == 9
    for c in golden_complex["linear"]:
        lin = orc.ComplexLinearOracle(c["W"])
        assert lin.rank == c["rank"]
        close(lin.n_eff(c["az"], c["ax"]), c["n_eff"], 1e-12)
        close(lin.backward_mean(c["az"], c["bz"], c["ax"], c["bx"]), c["rz"], 1e-10, 1e-12)
        close(lin.forward_mean(c["az"], c["bz"], c["ax"], c["bx"]), c["rx"], 1e-10, 1e-12)
        close(lin.backward_variance(c["az"], c["ax"]), c["vz"], 1e-12)
        close(lin.forward_variance(c["az"], c["ax"]), c["vx"], 1e-12)
        a, b = lin.backward_message(c["az"], c["bz"], c["ax"], c["bx"])
        close(a, c["a_bwd"], 1e-9); close(b, c["b_bwd"], 1e-8, 1e-10 * np.abs(c["b_bwd"]).max())
        a, b = lin.forward_message(c["az"], c["bz"], c["ax"], c["bx"])
        close(a, c["a_fwd"], 1e-9); close(b, c["b_fwd"], 1e-8, 1e-10 * np.abs(c["b_fwd"]).max())
        close(lin.log_partition(c["az"], c["bz"], c["ax"], c["bx"]), c["A"], 1e-10)
        # the real embedding [[Re W, -Im W], [Im W, Re W]] the HIP path factorises gives the same posterior
        Wr = np.block([[c["W"].real, -c["W"].imag], [c["W"].imag, c["W"].real]])
        emb = orc.LinearOracle(Wr)
        rz = emb.backward_mean(c["az"], c["bz"].reshape(-1), c["ax"], c["bx"].reshape(-1)).reshape(2, -1)
        close(rz, c["rz"], 1e-9, 1e-11)
        close(emb.backward_variance(c["az"], c["ax"]), c["vz"], 1e-11)
        close(emb.forward_variance(c["az"], c["ax"]), c["vx"], 1e-11)


def test_phase_retrieval_ep_trajectories(golden_complex):
    assert len(golden_complex["ep"]) == 3
    for c in golden_complex["ep"]:
        assert c["error"] is None
        W = redraw_complex_W(c)
        N, M = c["N"], c["M"]
        fs = [orc.GBPriorFactor((2, N), rho=c["rho"]), orc.ComplexLinearFactor(W), orc.ModulusLikFactor(c["y"])]
        recs, record = _recorder()

        def init(i, d):
            return (c["a0"], c["b0"]) if (i, d) == (0, "bwd") else (0, 0.0)
        with np.errstate(all="ignore"):
            res = orc.ep_chain(fs, [(2, N), (2, M)], max_iter=25, damping=c["damping"], record=record, init=init)
        assert res["n_iter"] == c["n_iter"]
        _check_traj(recs, c["traj"], ["x", "z"])


def test_modulus_se_error_and_phase_retrieval_se(golden_complex):
    assert len(golden_complex["se"]) == 10
    for c in golden_complex["se"]:
        v = orc.se_modulus_lik_error(c["az"], c["tau_z"])
        close(v, c["v"], 1e-9)
        close(orc.a_new(v, c["az"]), c["a_new"], 1e-7)
    for c in golden_complex["se_traj"][:1]:
        fs = [orc.GBPriorFactor(None, rho=c["rho"]), orc.MPFactor(c["alpha"]), orc.ModulusLikFactor(None)]
        recs = []
        res = orc.se_chain(fs, max_iter=60, a0=c["a0"], record=lambda it, f, b, v: recs.append((f, b, v)))
        assert res["n_iter"] == c["n_iter"]
        t = c["traj"]
        for it in range(len(recs)):
            for i, vid in enumerate(["x", "z"]):
                close(recs[it][0][i], t[f"fwd:{vid}:a"][it], 1e-5)
                close(recs[it][1][i], t[f"bwd:{vid}:a"][it], 1e-5)
                close(recs[it][2][i], t[f"v:{vid}"][it], 1e-5)
