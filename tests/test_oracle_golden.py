"""Pin the CPU oracle on golden vectors captured from the reference.

The goldens were produced by tools/make_golden.py importing sphinxteam/tramp in
the build container.  The oracle must reproduce them to round-off (it restates
the same NumPy/SciPy operations); masks must match exactly.
"""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

RT = 1e-12      # oracle vs reference, fp64


def close(a, b, rtol=RT, atol=0.0):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# -- SURVEY.md section 8(c) known-answer values (typed in from the survey, not from the npz)
def test_survey_known_answers():
    b = np.array([-3, -.5, 0, .25, 2, 6.])
    bx = np.array([.5, -1, 2, 0, -.3, 1.5])
    y = np.array([1, -1, 1, 1, -1, 1.])
    _, _, eta = orc.gb_params(.1, 0, 1)
    close(eta, 3.116163110540892)
    r, v = orc.gb_prior_posterior(1.3, b, rho=.1)
    close(r, [-0.4452762405163541, -0.01560915113229883, 0, 0.00751440085696626,
              0.12938360312113706, 2.5945590962882785])
    close(v, 0.20922982079683708)
    a_new, b_new = orc.gb_prior_message(1.3, b, rho=.1)
    close(a_new, 3.4794334296687266)
    close(b_new, [0.8718318506389249, 0.4253971012695395, 0, -0.21408542134028413,
                  -1.3816196819918467, 6.400522480051279])
    r, v = orc.l1_prior_posterior(1.3, b, gamma=1)
    close(r, [-1.5384615384615383, -0, 0, 0, 0.7692307692307692, 3.846153846153846])
    close(v, 0.3846153846153846)
    assert list(orc.l1_support_mask(b, 1)) == [True, False, False, False, True, True]
    r, v = orc.sgn_lik_posterior(1.3, b, y)
    close(r, [0.2731444915438206, -0.8593267684716833, 0.69979105253342, 0.7744441511213647,
              -0.3536398698019141, 4.615384954720828])
    close(v, 0.3154745509879305)
    close(orc.sgn_lik_message(1.3, b, y)[0], 1.869827793932761)
    for kind, vf, vb in [("relu", 0.2744553290012942, 0.6363319425833475),
                         ("abs", 0.34089661751161554, 0.6474879337683331),
                         ("sgn", 0.26355799942525987, 0.6659266259870412)]:
        close(orc.pwl_channel_posterior(kind, "fwd", 1.3, b, 0.7, bx)[1], vf)
        close(orc.pwl_channel_posterior(kind, "bwd", 1.3, b, 0.7, bx)[1], vb)
    close(orc.se_gb_prior_error(1.3, rho=.1), 0.0799760619557935, rtol=1e-9)
    close(orc.se_sgn_lik_error(1.3, 1.0), 0.3262355943975024, rtol=1e-9)


def test_gb_prior(golden_factors):
    for c in golden_factors["gb"]:
        _, _, eta = orc.gb_params(c["rho"], c["mean"], c["var"])
        close(eta, c["eta"])
        r, v = orc.gb_prior_posterior(c["ax"], c["b"], c["rho"], c["mean"], c["var"])
        close(r, c["r"]); close(v, c["v"])
        a, b = orc.gb_prior_message(c["ax"], c["b"], c["rho"], c["mean"], c["var"])
        close(a, c["a_new"]); close(b, c["b_new"])
        _, vd = orc.gb_prior_posterior(c["ax"], c["b"], c["rho"], c["mean"], c["var"],
                                       isotropic=False)
        close(vd, c["v_diag"])
    d = golden_factors["gb_diag"]
    r, v = orc.gb_prior_posterior(d["ax"], d["bx"], 0.5, isotropic=False)
    close(r, d["r"]); close(v, d["v"])
    a, b = orc.gb_prior_message(d["ax"], d["bx"], 0.5, isotropic=False)
    close(a, d["a_new"]); close(b, d["b_new"])


def test_l1_prior_and_mask(golden_factors):
    for c in golden_factors["l1"]:
        with np.errstate(all="ignore"):
            r, v = orc.l1_prior_posterior(c["ax"], c["b"], c["gamma"])
            a, b = orc.l1_prior_message(c["ax"], c["b"], c["gamma"])
        close(r, c["r"]); close(v, c["v"]); close(a, c["a_new"]); close(b, c["b_new"])
        assert np.array_equal(orc.l1_support_mask(c["b"], c["gamma"]), c["mask"])


def test_gaussian_prior_likelihood(golden_factors):
    c = golden_factors["gauss_prior"]
    a, b = orc.gauss_prior_message(c["ax"], c["b"], c["mean"], c["var"])
    close(a, c["a_new"]); close(b, c["b_new"])
    r, v = orc.gauss_prior_posterior(c["ax"], c["b"], c["mean"], c["var"])
    close(r, c["r"]); close(v, c["v"])
    close(orc.gauss_prior_error(c["ax"], c["var"]), c["err"])
    c = golden_factors["gauss_lik"]
    a, b = orc.gauss_lik_message(c["az"], c["b"], c["y"], c["var"])
    close(a, c["a_new"]); close(b, c["b_new"])
    r, v = orc.gauss_lik_posterior(c["az"], c["b"], c["y"], c["var"])
    close(r, c["r"]); close(v, c["v"])
    close(orc.gauss_lik_error(c["az"], 1.0, c["var"]), c["err"])


def test_sgn_abs_likelihood(golden_factors):
    for c in golden_factors["sgn_lik"]:
        with np.errstate(all="ignore"):
            r, v = orc.sgn_lik_posterior(c["az"], c["bz"], c["y"])
            a, b = orc.sgn_lik_message(c["az"], c["bz"], c["y"])
            _, vd = orc.sgn_lik_posterior(c["az"], c["bz"], c["y"], isotropic=False)
        close(r, c["r"]); close(v, c["v"]); close(a, c["a_new"]); close(b, c["b_new"])
        close(vd, c["v_diag"])
    for c in golden_factors["abs_lik"]:
        r, v = orc.abs_lik_posterior(c["az"], c["bz"], c["y"])
        a, b = orc.abs_lik_message(c["az"], c["bz"], c["y"])
        close(r, c["r"]); close(v, c["v"]); close(a, c["a_new"]); close(b, c["b_new"])


def test_pwl_channels(golden_factors):
    for c in golden_factors["pwl"]:
        with np.errstate(all="ignore"):
            for d in ("fwd", "bwd"):
                r, v = orc.pwl_channel_posterior(c["kind"], d, c["az"], c["bz"], c["ax"], c["bx"])
                a, b = orc.pwl_channel_message(c["kind"], d, c["az"], c["bz"], c["ax"], c["bx"])
                np.testing.assert_allclose(r, c["r_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(v, c["v_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(a, c["a_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(b, c["b_" + d], rtol=RT, equal_nan=True)


def test_pwl2_channels(golden_factors):
    "LeakyRelu / AsymmetricAbs (generic two-half-line form)"
    for c in golden_factors["pwl2"]:
        with np.errstate(all="ignore"):
            for d in ("fwd", "bwd"):
                r, v = orc.pwl_channel_posterior(c["kind"], d, c["az"], c["bz"], c["ax"], c["bx"], param=c["param"])
                a, b = orc.pwl_channel_message(c["kind"], d, c["az"], c["bz"], c["ax"], c["bx"], param=c["param"])
                np.testing.assert_allclose(r, c["r_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(v, c["v_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(a, c["a_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(b, c["b_" + d], rtol=RT, equal_nan=True)
        close(orc.pwl_second_moment(c["kind"], 1.7, c["param"]), c["tau_x"], rtol=1e-12)


def test_pwl3_channels(golden_factors):
    "HardTanh / HardSigmoid / SymmetricDoor: finite-interval truncated normal incl. branch coverage"
    seen = np.zeros(4, dtype=int)
    for c in golden_factors["pwl3"]:
        with np.errstate(all="ignore"):
            for d in ("fwd", "bwd"):
                r, v = orc.pwl3_channel_posterior(c["kind"], d, c["az"], c["bz"], c["ax"], c["bx"], param=c["param"])
                a, b = orc.pwl3_channel_message(c["kind"], d, c["az"], c["bz"], c["ax"], c["bx"], param=c["param"])
                np.testing.assert_allclose(r, c["r_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(v, c["v_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(a, c["a_" + d], rtol=RT, equal_nan=True)
                np.testing.assert_allclose(b, c["b_" + d], rtol=RT, equal_nan=True)
            seen += [int(m.sum()) for m in orc.pwl3_mid_masks(c["kind"], c["az"], c["bz"], c["ax"], c["bx"], c["param"])]
        close(orc.pwl3_second_moment(c["kind"], 1.7, c["param"]), c["tau_x"], rtol=1e-12)
    assert seen[1] > 0 and seen[2] > 0 and seen[3] > 0      # neg, pos, other branches all exercised


def test_relu_likelihoods(golden_factors):
    for c in golden_factors["relu_lik"]:
        with np.errstate(all="ignore"):
            r, v = orc.relu_lik_posterior(c["az"], c["bz"], c["y"], c["slope"])
            a, b = orc.relu_lik_message(c["az"], c["bz"], c["y"], c["slope"])
        np.testing.assert_allclose(r, c["r"], rtol=RT, equal_nan=True)
        np.testing.assert_allclose(v, c["v"], rtol=RT, equal_nan=True)
        np.testing.assert_allclose(a, c["a_new"], rtol=RT, equal_nan=True)
        np.testing.assert_allclose(b, c["b_new"], rtol=1e-10, equal_nan=True)


def test_log_Phi_mask(golden_factors):
    c = golden_factors["log_Phi"]
    close(orc.log_Phi(c["x"]), c["y"])
    assert np.array_equal(c["x"] < 30, c["mask"])
    assert np.all(orc.log_Phi(c["x"][~c["mask"]]) == 0.0)


def _W(case):
    if "W" in case:
        return case["W"]
    rs = np.random.RandomState(case["seed"])
    W = rs.randn(case["M"], case["N"]) / np.sqrt(case["N"])
    assert W.sum() == case["W_sum"]
    return W


@pytest.mark.parametrize("thin", [False, True])
def test_linear_channel(golden_linear, thin):
    rt = 1e-9 if thin else 1e-10     # thin form re-associates the sums
    for case in golden_linear:
        lin = (orc.LinearThinOracle if thin else orc.LinearOracle)(_W(case))
        assert lin.rank == case["rank"]
        close(lin.spectrum, case["spectrum"], rtol=1e-10, atol=1e-14)
        bz, bx = case["bz"], case["bx"]
        scale = max(np.abs(bz).max(), np.abs(bx).max())
        for c in case["cases"]:
            az, ax = c["az"], c["ax"]
            rz = lin.backward_mean(az, bz, ax, bx)
            rx = lin.forward_mean(az, bz, ax, bx)
            close(rz, c["rz"], rtol=rt, atol=rt * np.abs(c["rz"]).max())
            # the reference forms rx = W @ rz; for tiny az the null-space part of rz is
            # ~|bz|/az and cancels in W @ rz, so the REFERENCE value carries that round-off
            cancel = 2e-16 * np.abs(bz).max() / max(az, 1e-11) * case["N"]
            close(rx, c["rx"], rtol=rt, atol=rt * np.abs(c["rx"]).max() + cancel)
            close(lin.backward_variance(az, ax), c["vz"], rtol=1e-11)
            close(lin.forward_variance(az, ax), c["vx"], rtol=1e-11)
            close(lin.n_eff(az, ax), c["n_eff"], rtol=1e-11)
            if "rzK" in c:
                close(lin.backward_mean(az, case["bzK"], ax, case["bxK"]), c["rzK"],
                      rtol=rt, atol=rt * np.abs(c["rzK"]).max())
            if "a_fwd" in c:
                a, b = lin.forward_message(az, bz, ax, bx)
                close(a, c["a_fwd"], rtol=1e-10)
                close(b, c["b_fwd"], rtol=1e-8,
                      atol=1e-9 * np.abs(c["b_fwd"]).max() + cancel * (ax + a))
                a, b = lin.backward_message(az, bz, ax, bx)
                close(a, c["a_bwd"], rtol=1e-10)
                close(b, c["b_bwd"], rtol=1e-8, atol=1e-9 * np.abs(c["b_bwd"]).max())
            # algebraic identity (precompute_svd=False branch of the reference)
            if 1e-6 < az < 1e6 and ax < 1e6:
                close(rz, lin.backward_mean_solve(az, bz, ax, bx), rtol=1e-8,
                      atol=1e-9 * scale)
        if "vz_az0" in case:
            close(lin.backward_variance(0., 0.7), case["vz_az0"], rtol=1e-11)
            close(lin.second_moment(1.7), case["second_moment"], rtol=1e-11)


def _glm_factors(c):
    N, M = c["N"], int(c["alpha"] * c["N"])
    rs = np.random.RandomState(c["seed"])
    W = rs.randn(M, N) / np.sqrt(N)
    assert W.sum() == c["W_sum"]
    prior = orc.GBPriorFactor(N, rho=c["rho"])
    lik = {"gaussian": lambda: orc.GaussLikFactor(c["y"], c["noise_var"]),
           "sgn": lambda: orc.SgnLikFactor(c["y"]),
           "abs": lambda: orc.AbsLikFactor(c["y"])}[c["output"]]()
    return [prior, orc.LinearFactor(W), lik], [N, M], ["x", "z"]


def _two_layer_factors(c):
    N, M1, M2 = c["N"], c["M1"], c["M2"]
    rs = np.random.RandomState(c["seed"])
    W1 = rs.randn(M1, N) / np.sqrt(N)
    W2 = rs.randn(M2, M1) / np.sqrt(M1)
    assert W1.sum() == c["W1_sum"] and W2.sum() == c["W2_sum"]
    fs = [orc.GaussPriorFactor(N), orc.LinearFactor(W1), orc.PWLFactor("relu"),
          orc.LinearFactor(W2), orc.GaussLikFactor(c["y"], c["noise_var"])]
    return fs, [N, M1, M1, M2], ["x", "z1", "a1", "z2"]


def test_ep_trajectories(golden_ep):
    """Row H: the oracle's chain scheduler reproduces the reference's iterate()."""
    for c in golden_ep:
        fs, shapes, ids = _glm_factors(c) if c["kind"] == "glm" else _two_layer_factors(c)
        recs = []

        def record(it, fwd, bwd, rs_, vs_):
            recs.append(dict(fwd=[(a, b.copy()) for a, b in fwd],
                             bwd=[(a, np.array(b, dtype=float).copy()) for a, b in bwd],
                             r=[r.copy() for r in rs_], v=list(vs_)))
        with np.errstate(all="ignore"):
            res = orc.ep_chain(fs, shapes, max_iter=30, damping=c.get("damping"), record=record)
        assert res["n_iter"] == c["n_iter"], (c["kind"], c.get("output"), c["N"], c["seed"])
        t = c["traj"]
        n = t[f"v:{ids[0]}"].shape[0]
        for it in range(n):
            for i, vid in enumerate(ids):
                tol = 1e-7 if it < 6 else 1e-5      # round-off amplifies along the trajectory
                close(recs[it]["fwd"][i][0], t[f"fwd:{vid}:a"][it], rtol=tol)
                close(recs[it]["bwd"][i][0], t[f"bwd:{vid}:a"][it], rtol=tol)
                sc = np.abs(t[f"fwd:{vid}:b"][it]).max()
                close(recs[it]["fwd"][i][1], t[f"fwd:{vid}:b"][it], rtol=tol, atol=tol * sc)
                sc = np.abs(t[f"bwd:{vid}:b"][it]).max()
                close(recs[it]["bwd"][i][1], t[f"bwd:{vid}:b"][it], rtol=tol, atol=tol * sc)
                sc = np.abs(t[f"r:{vid}"][it]).max()
                close(recs[it]["r"][i], t[f"r:{vid}"][it], rtol=tol, atol=tol * sc)
                close(recs[it]["v"][i], t[f"v:{vid}"][it], rtol=tol)


def test_diagonal_beliefs_oracle_matches_reference(golden_diag):
    """isotropic=False forms (every factor-level test of the reference runs in this mode: tests/test_priors.py:17-51,
    tests/test_likelihoods.py:22-83): posteriors as arrays, messages with the element-wise clip of base.py:250-255;
    the Gaussian factors answer with their constant message whatever comes in (gaussian_prior.py:86-89,
    gaussian_likelihood.py:68-71)."""
    from tests.conftest import diag_oracle_posterior
    assert len(golden_diag) == 44
    for c in golden_diag:
        with np.errstate(all="ignore"):
            r, v = diag_oracle_posterior(orc, c)
        tag = (c["family"], c["name"], c["inputs"], c["kwargs"])
        np.testing.assert_allclose(r, c["r"], rtol=1e-14, atol=1e-15, err_msg=str(tag))
        np.testing.assert_allclose(np.broadcast_to(v, c["v"].shape), c["v"], rtol=1e-13, atol=1e-16, err_msg=str(tag))
        if c["name"] == "gaussian":
            continue
        an, bn = orc.ab_new(r, v, c["a"], c["b"])
        np.testing.assert_allclose(an, c["a_new"], rtol=1e-11, err_msg=str(tag))
        np.testing.assert_allclose(bn, c["b_new"], rtol=1e-11, atol=1e-12 * np.abs(c["b_new"]).max(), err_msg=str(tag))


@pytest.mark.parametrize("damping", [None, 0.1])
@pytest.mark.parametrize("M,N", [(100, 200), (120, 120), (90, 60)])
def test_batched_thin_ep_matches_the_chain_scheduler(M, N, damping):
    """ep_glm_thin_batch (the host counterpart of the ensemble run, used by bench.py's like-for-like CPU baseline)
    against ep_chain - itself pinned on the reference's trajectories above - instance by instance."""
    rs = np.random.RandomState(M + N)
    B, rho, nv = 3, 0.1, 1e-2
    W = orc.gaussian_ensemble(rs, M, N)
    X = np.stack([orc.gb_sample(rs, N, rho) for _ in range(B)])
    Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((B, M))
    lin = orc.LinearFactor(W, thin=True)
    L = lin.lin
    res = orc.ep_glm_thin_batch(L.Uk, L.sigma, L.Vk.T, Y, rho, nv, damping=damping, n_iter=8, rank=L.rank)
    for b in range(B):
        ref = orc.ep_chain([orc.GBPriorFactor(N, rho=rho), lin, orc.GaussLikFactor(Y[b], nv)], [N, M], max_iter=8,
                           damping=damping, tol=0.0, max_increase=np.inf)
        for i in range(2):
            close(res["r"][i][b], ref["r"][i], rtol=1e-9, atol=1e-11)
            close(res["v"][i][b], ref["v"][i], rtol=1e-10)
            close(res["fwd"][i][0][b], ref["fwd"][i][0], rtol=1e-9)
            close(res["bwd"][i][0][b], ref["bwd"][i][0], rtol=1e-9)
            sc = np.abs(ref["bwd"][i][1]).max()
            close(res["bwd"][i][1][b], ref["bwd"][i][1], rtol=1e-9, atol=1e-10 * sc)


def test_se_scalar_rows(golden_se):
    for c in golden_se["gb_prior"]:
        v = orc.se_gb_prior_error(c["ax"], c["rho"], c["mean"], c["var"])
        close(v, c["v"], rtol=1e-9)
        v = orc.se_gb_prior_error(c["ax"], c["rho"], c["mean"], c["var"], method="gl")
        close(v, c["v"], rtol=2e-7, atol=1e-9)
    for c in golden_se["sgn_lik"]:
        close(orc.se_sgn_lik_error(c["az"], c["tau_z"]), c["v"], rtol=1e-9)
        close(orc.se_sgn_lik_error(c["az"], c["tau_z"], method="gl"), c["v"], rtol=2e-7)
    for c in golden_se.get("relu_lik", []):
        close(orc.se_relu_lik_error(c["az"], c["tau_z"]), c["v"], rtol=1e-9)
        close(orc.se_relu_lik_error(c["az"], c["tau_z"], method="gl"), c["v"], rtol=2e-7)
    for c in golden_se["mp"]:
        mp = orc.MarchenkoPasturOracle(c["alpha"])
        close(mp.mean_spectrum, c["mean_spectrum"], rtol=1e-12)
        close(mp.backward_error(c["az"], c["ax"]), c["vz"], rtol=1e-12)
        close(mp.forward_error(c["az"], c["ax"]), c["vx"], rtol=1e-12)
        close(mp.second_moment(1.7), c["tau_x"], rtol=1e-12)


def test_se_trajectories(golden_se):
    for c in golden_se["traj"]:
        lik = (orc.GaussLikFactor(None, c["noise_var"]) if c["kind"] == "glm_gauss"
               else orc.SgnLikFactor(None))
        fs = [orc.GBPriorFactor(None, rho=c["rho"]), orc.MPFactor(c["alpha"]), lik]
        recs = []
        res = orc.se_chain(fs, max_iter=200, a0=c["a0"],
                           record=lambda it, f, b, v: recs.append((f, b, v)))
        assert res["n_iter"] == c["n_iter"]
        t = c["traj"]
        for it in range(len(recs)):
            for i, vid in enumerate(["x", "z"]):
                close(recs[it][0][i], t[f"fwd:{vid}:a"][it], rtol=1e-7)
                close(recs[it][1][i], t[f"bwd:{vid}:a"][it], rtol=1e-7)
                close(recs[it][2][i], t[f"v:{vid}"][it], rtol=1e-7)


def test_se_2d_rows_gl(golden_se):
    """2-D rows: the fixed Gauss-Legendre form against the reference's dblquad values.

    (The quadpack form of the oracle is the same call sequence as the reference
    and takes seconds per value; it is exercised in the slow test below.)
    """
    if "pwl" not in golden_se:
        pytest.skip("slow goldens not generated")
    for c in golden_se["pwl"]:
        with np.errstate(all="ignore"):
            vf = orc.se_pwl_channel_error(c["kind"], "fwd", c["az"], c["ax"], c["tau_z"], "gl")
            vb = orc.se_pwl_channel_error(c["kind"], "bwd", c["az"], c["ax"], c["tau_z"], "gl")
        close(vf, c["v_fwd"], rtol=1e-6, atol=1e-8)
        close(vb, c["v_bwd"], rtol=1e-6, atol=1e-8)
        close(orc.pwl_second_moment(c["kind"], c["tau_z"]), c["tau_x"], rtol=1e-12)
    for c in golden_se["abs_lik"]:
        close(orc.se_abs_lik_error(c["az"], c["tau_z"], "gl"), c["v"], rtol=1e-6)
    for c in golden_se.get("pwl3", []):
        with np.errstate(all="ignore"):
            vf = orc.se_pwl3_channel_error(c["kind"], "fwd", c["az"], c["ax"], c["tau_z"], "gl", c["param"])
            vb = orc.se_pwl3_channel_error(c["kind"], "bwd", c["az"], c["ax"], c["tau_z"], "gl", c["param"])
        close(vf, c["v_fwd"], rtol=1e-6, atol=1e-8)
        close(vb, c["v_bwd"], rtol=1e-6, atol=1e-8)
        close(orc.pwl3_second_moment(c["kind"], c["tau_z"], c["param"]), c["tau_x"], rtol=1e-12)
    for c in golden_se.get("pwl2", []):
        with np.errstate(all="ignore"):
            vf = orc.se_pwl_channel_error(c["kind"], "fwd", c["az"], c["ax"], c["tau_z"], "gl", c["param"])
            vb = orc.se_pwl_channel_error(c["kind"], "bwd", c["az"], c["ax"], c["tau_z"], "gl", c["param"])
        close(vf, c["v_fwd"], rtol=1e-6, atol=1e-8)
        close(vb, c["v_bwd"], rtol=1e-6, atol=1e-8)
        close(orc.pwl_second_moment(c["kind"], c["tau_z"], c["param"]), c["tau_x"], rtol=1e-12)


@pytest.mark.slow
def test_se_2d_rows_quadpack_one(golden_se):
    if "pwl" not in golden_se:
        pytest.skip("slow goldens not generated")
    c = [c for c in golden_se["pwl"] if c["kind"] == "relu"][0]
    with np.errstate(all="ignore"):
        vf = orc.se_pwl_channel_error("relu", "fwd", c["az"], c["ax"], c["tau_z"])
    close(vf, c["v_fwd"], rtol=1e-9)


# -- convolution channel family
def conv_filter_of(c):
    "Rebuild the filter from the constructor arguments (checks the filter builders too)."
    kw = c["kwargs"]
    if c["kind"] == "conv":
        return np.asarray(kw["filter"])
    if c["kind"] == "blur_1d":
        return orc.gaussian_filter(kw["sigma"], kw["N"])
    if c["kind"] == "blur_2d":
        return np.outer(orc.gaussian_filter(kw["sigma"][0], kw["shape"][0]),
                        orc.gaussian_filter(kw["sigma"][1], kw["shape"][1]))
    if c["kind"] == "laplacian":
        return orc.stencil_filter(tuple(kw["shape"]), None, [1.0] * len(kw["shape"]))
    return orc.stencil_filter(tuple(kw["shape"]), kw["D1"], kw["D2"])


def test_conv_channel_oracle_matches_reference(golden_conv):
    for c in golden_conv["channels"]:
        f = conv_filter_of(c)
        close(f, c["filter"], rtol=1e-12, atol=1e-15)
        o = orc.ConvOracle(c["filter"])
        close(o.spectrum, c["spectrum"], rtol=1e-12, atol=1e-14)
        close(o.second_moment(1.7), c["second_moment"])
        close(o.convolve(c["z"]), c["x"], rtol=1e-12, atol=1e-13)
        for k in c["cases"]:
            az, ax = k["az"], k["ax"]
            with np.errstate(all="ignore"):
                sc = np.abs(k["rz"]).max()
                close(o.backward_mean(az, c["bz"], ax, c["bx"]), k["rz"], rtol=1e-11, atol=1e-12 * sc)
                close(o.forward_mean(az, c["bz"], ax, c["bx"]), k["rx"], rtol=1e-11,
                      atol=1e-12 * np.abs(k["rx"]).max())
                close(o.n_eff(az, ax), k["n_eff"])
                close(o.backward_variance(az, ax), k["vz"])
                close(o.forward_variance(az, ax), k["vx"])
                a, b = o.forward_message(az, c["bz"], ax, c["bx"])
                close(a, k["a_fwd"])
                close(b, k["b_fwd"], rtol=1e-10, atol=1e-11 * np.abs(k["b_fwd"]).max())
                a, b = o.backward_message(az, c["bz"], ax, c["bx"])
                close(a, k["a_bwd"])
                close(b, k["b_bwd"], rtol=1e-10, atol=1e-11 * np.abs(k["b_bwd"]).max())


def test_conv_ep_oracle_matches_reference(golden_conv):
    for c in golden_conv["ep"]:
        shape = tuple(c["shape"])
        fs = [orc.GBPriorFactor(shape, rho=c["rho"]), orc.ConvFactor(c["filter"]),
              orc.GaussLikFactor(c["y"], c["noise_var"])]
        iters = []
        res = orc.ep_chain(fs, [shape, shape], max_iter=40, damping=c["damping"],
                           record=lambda it, fwd, bwd, rs, vs: iters.append((list(fwd), list(bwd), rs, vs)))
        assert res["n_iter"] == c["n_iter"], c["name"]
        t = c["traj"]
        for it in range(min(len(iters), t["v:x"].shape[0])):
            fwd, bwd, rs, vs = iters[it]
            for i, vid in enumerate(["x", "z"]):
                close(fwd[i][0], t[f"fwd:{vid}:a"][it], rtol=1e-8)
                close(bwd[i][0], t[f"bwd:{vid}:a"][it], rtol=1e-8)
                close(rs[i], t[f"r:{vid}"][it], rtol=1e-7, atol=1e-9 * np.abs(t[f"r:{vid}"][it]).max())
                close(vs[i], t[f"v:{vid}"][it], rtol=1e-8)
        close(res["r"][0], c["final_r"], rtol=1e-6, atol=1e-7)


# -- three-region likelihoods (likelihood twins)
def test_pwl3_lik_oracle_matches_reference(golden_pwl3_lik):
    for c in golden_pwl3_lik["cases"]:
        with np.errstate(all="ignore"):
            r, v = orc.pwl3_lik_posterior(c["kind"], c["az"], c["bz"], c["y"], param=c["param"])
            a, b = orc.pwl3_lik_message(c["kind"], c["az"], c["bz"], c["y"], param=c["param"])
        close(r, c["r"], rtol=1e-10, atol=1e-12)
        close(v, c["v"], rtol=1e-10)
        close(a, c["a_new"], rtol=1e-9)
        close(b, c["b_new"], rtol=1e-9, atol=1e-10 * np.abs(c["b_new"]).max())
    for c in golden_pwl3_lik["se"]:
        close(orc.se_pwl3_lik_error(c["kind"], c["az"], c["tau_z"], param=c["param"]), c["v"], rtol=1e-7)
        close(orc.se_pwl3_lik_error(c["kind"], c["az"], c["tau_z"], method="gl", param=c["param"]), c["v"],
              rtol=2e-6)


def test_pwl3_lik_ep_oracle_matches_reference(golden_pwl3_lik):
    for c in golden_pwl3_lik["ep"]:
        N, M = c["N"], c["M"]
        W = np.random.RandomState(c["seed"]).randn(M, N) / np.sqrt(N)
        assert W.sum() == c["W_sum"]
        fs = [orc.GaussPriorFactor(N), orc.LinearFactor(W), orc.Pwl3LikFactor(c["kind"], c["y"], c["param"])]
        iters = []
        with np.errstate(all="ignore"):
            res = orc.ep_chain(fs, [N, M], max_iter=30, damping=c["damping"],
                               record=lambda it, fwd, bwd, rs, vs: iters.append((list(fwd), list(bwd), rs, vs)))
        assert res["n_iter"] == c["n_iter"]
        t = c["traj"]
        for it in range(min(len(iters), t["v:x"].shape[0])):
            fwd, bwd, rs, vs = iters[it]
            for i, vid in enumerate(["x", "z"]):
                close(fwd[i][0], t[f"fwd:{vid}:a"][it], rtol=1e-7)
                close(bwd[i][0], t[f"bwd:{vid}:a"][it], rtol=1e-7)
                close(rs[i], t[f"r:{vid}"][it], rtol=1e-6, atol=1e-8 * np.abs(t[f"r:{vid}"][it]).max())
                close(vs[i], t[f"v:{vid}"][it], rtol=1e-7)


# -- unitary channels (DFT, rotation)
def _unitary_factor(c):
    return orc.DFTFactor() if c["kind"] == "dft" else orc.RotationFactor(c["R"])


def test_unitary_channels_oracle_matches_reference(golden_unitary):
    for c in golden_unitary["msgs"]:
        f = _unitary_factor(c)
        a, b = f.forward(1.3, c["bz"], 0.7, c["bx"])
        assert a == c["a_fwd"]
        close(b, c["b_fwd"], rtol=1e-12, atol=1e-13)
        a, b = f.backward(1.3, c["bz"], 0.7, c["bx"])
        assert a == c["a_bwd"]
        close(b, c["b_bwd"], rtol=1e-12, atol=1e-13)
    for c in golden_unitary["ep"]:
        N = c["N"]
        zshape = (2, N) if c["kind"] == "dft" else (N,)
        fs = [orc.GBPriorFactor(N, rho=c["rho"]), _unitary_factor(c), orc.GaussLikFactor(c["y"], c["noise_var"])]
        iters = []
        res = orc.ep_chain(fs, [(N,), zshape], max_iter=30, damping=c["damping"],
                           record=lambda it, fwd, bwd, rs, vs: iters.append((rs, vs)))
        assert res["n_iter"] == c["n_iter"]
        t = c["traj"]
        for it in range(min(len(iters), t["v:x"].shape[0])):
            for i, vid in enumerate(["x", "z"]):
                close(iters[it][0][i], t[f"r:{vid}"][it], rtol=1e-8, atol=1e-10)
                close(iters[it][1][i], t[f"v:{vid}"][it], rtol=1e-9)


# -- log-partitions and log-evidence
def test_log_partitions_oracle_matches_reference(golden_logz):
    for c in golden_logz["cases"]:
        az, ax, bz, bx = c["az"], c["ax"], c["bz"], c["bx"]
        close(orc.gb_prior_log_partition(ax, bx, 0.1, 0.3, 1.7), c["gb"])
        close(orc.gauss_prior_log_partition(ax, bx, 0.3, 1.7), c["gauss_prior"])
        close(orc.l1_prior_log_partition(ax, bx, 0.8), c["l1"])
        close(orc.gauss_lik_log_partition(az, bz, c["y_g"], 0.4), c["gauss_lik"])
        close(orc.sgn_lik_log_partition(az, bz, c["y_sgn"]), c["sgn_lik"])
        close(orc.abs_lik_log_partition(az, bz, c["y_abs"]), c["abs_lik"])
        for kind in ("sgn", "abs", "relu"):
            close(orc.pwl_channel_log_partition(kind, az, bz, ax, bx), c["pwl_" + kind])
        close(orc.variable_log_partition(az + ax, bz + bx), c["variable"])
    for c in golden_logz["lin"]:
        lin = orc.LinearOracle(c["W"])
        for k in c["cases"]:
            close(orc.linear_log_partition(lin, k["az"], c["bz"], k["ax"], c["bx"]), k["logZ"], rtol=1e-10)


def _logz_chain(c):
    if c["kind"] == "glm":
        N, M = c["N"], int(c["alpha"] * c["N"])
        W = np.random.RandomState(c["seed"]).randn(M, N) / np.sqrt(N)
        assert W.sum() == c["W_sum"]
        lik = {"gaussian": lambda: orc.GaussLikFactor(c["y"], c["noise_var"]), "sgn": lambda: orc.SgnLikFactor(c["y"]),
               "abs": lambda: orc.AbsLikFactor(c["y"])}[c["output"]]()
        return [orc.GBPriorFactor(N, rho=c["rho"]), orc.LinearFactor(W), lik], [N, M]
    fs = [orc.GaussPriorFactor(c["N"]), orc.LinearFactor(c["W1"]), orc.PWLFactor("relu"), orc.LinearFactor(c["W2"]),
          orc.GaussLikFactor(c["y"], c["noise_var"])]
    return fs, [c["N"], c["M1"], c["M1"], c["M2"]]


def test_ep_log_evidence_oracle_matches_reference(golden_logz):
    for c in golden_logz["ep"]:
        fs, shapes = _logz_chain(c)
        As = []
        with np.errstate(all="ignore"):
            orc.ep_chain(fs, shapes, max_iter=len(c["A"]), damping=c["damping"], tol=0.0, max_increase=np.inf,
                         record=lambda it, fwd, bwd, rs, vs: As.append(orc.chain_log_evidence(fs, fwd, bwd)))
        close(As, c["A"], rtol=1e-7, atol=1e-8 * np.abs(c["A"]).max())


# ---------------------------------------------------------------- trees (gradient channel, multi-factor variables)
TREE_EDGES = [("GaussianPrior", "x"), ("x", "GaussianLikelihood"), ("x", "GradientChannel"),
              ("GradientChannel", "z"), ("GaussBernoulliPrior", "z")]


def tree_factors(e):
    shape = tuple(e["shape"])
    d = len(shape)
    return dict(GaussianPrior=orc.GaussPriorFactor(shape), GaussianLikelihood=orc.GaussLikFactor(e["y"], e["noise_var"]),
                GradientChannel=orc.GradientFactor(shape),
                GaussBernoulliPrior=orc.GBPriorFactor((d,) + shape, rho=e["rho"])), dict(x=shape, z=(d,) + shape)


def test_gradient_unitary_oracle_matches_reference(golden_tree):
    for g in golden_tree["grads"]:
        o = orc.GradientOracle(g["shape"])
        np.testing.assert_allclose(o.spectrum, g["spectrum"], atol=1e-13)
        np.testing.assert_allclose(o.convolve(g["z"]), g["x"], atol=1e-13)
        assert abs(o.second_moment(1.7) - g["second_moment"]) < 1e-13
        for c in g["cases"]:
            with np.errstate(all="ignore"):
                fa, fb = o.forward_message(c["az"], g["bz"], c["ax"], g["bx"])
                ba, bb = o.backward_message(c["az"], g["bz"], c["ax"], g["bx"])
                assert abs(o.n_eff(c["az"], c["ax"]) - c["n_eff"]) < 1e-13
            np.testing.assert_allclose(fa, c["a_fwd"], rtol=1e-10)
            np.testing.assert_allclose(fb, c["b_fwd"], rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(ba, c["a_bwd"], rtol=1e-10)
            np.testing.assert_allclose(bb, c["b_bwd"], rtol=1e-9, atol=1e-10)
    for u in golden_tree["unit"]:
        f = orc.UnitaryFactor(u["U"])
        a, b = f.forward(1.3, u["bz"], 0.7, u["bx"])
        assert a == u["a_fwd"]
        np.testing.assert_allclose(b, u["b_fwd"], atol=1e-13)
        a, b = f.backward(1.3, u["bz"], 0.7, u["bx"])
        assert a == u["a_bwd"]
        np.testing.assert_allclose(b, u["b_bwd"], atol=1e-13)


def test_tree_ep_oracle_matches_reference(golden_tree):
    "every message of both kinds (factor -> variable, variable -> factor) and every posterior, per iteration"
    for e in golden_tree["ep"]:
        factors, shapes = tree_factors(e)
        recs = []

        def record(it, msg, rs, vs):
            recs.append(({k: (v[0], v[1].copy()) for k, v in msg.items()}, dict(rs), dict(vs)))
        orc.ep_tree(factors, ["x", "z"], TREE_EDGES, shapes, max_iter=12, damping=e["damping"], record=record)
        tr = e["traj"]
        assert len(recs) == 12
        for it, (msg, rs, vs) in enumerate(recs):
            for (s, t), (a, b) in msg.items():
                np.testing.assert_allclose(a, tr[f"{s}->{t}:a"][it], rtol=1e-12)
                np.testing.assert_allclose(b, tr[f"{s}->{t}:b"][it], rtol=1e-11, atol=1e-12)
            for v in ("x", "z"):
                np.testing.assert_allclose(rs[v], tr["r:" + v][it], rtol=1e-11, atol=1e-12)
                np.testing.assert_allclose(vs[v], tr["v:" + v][it], rtol=1e-12)


REG_EDGES = [("GaussianPrior", "x"), ("x", "GradientChannel"), ("GradientChannel", "x'"), ("GaussBernoulliPrior", "x'"),
             ("x", "ReshapeChannel"), ("ReshapeChannel", "r"), ("r", "LinearChannel"), ("LinearChannel", "z"),
             ("z", "GaussianLikelihood")]


def reg_factors(e):
    shape = tuple(e["shape"])
    d, N = len(shape), int(np.prod(shape))
    factors = dict(GaussianPrior=orc.GaussPriorFactor(shape), GradientChannel=orc.GradientFactor(shape),
                   GaussBernoulliPrior=orc.GBPriorFactor((d,) + shape, rho=e["rho"]),
                   ReshapeChannel=orc.ReshapeFactor(shape, (N,)), LinearChannel=orc.LinearFactor(e["A"]),
                   GaussianLikelihood=orc.GaussLikFactor(e["y"], e["noise_var"]))
    return factors, {"x": shape, "x'": (d,) + shape, "r": (N,), "z": (e["M"],)}


def test_tree_regression_oracle_matches_reference(golden_tree):
    "sparse-gradient regression: the gradient tree feeding a dense LinearChannel (total_variation_model.py:67-71)"
    for e in golden_tree["reg"]:
        factors, shapes = reg_factors(e)
        recs = []

        def record(it, msg, rs, vs):
            recs.append(({k: (v[0], v[1].copy()) for k, v in msg.items()}, dict(rs), dict(vs)))
        res = orc.ep_tree(factors, ["x", "x'", "r", "z"], REG_EDGES, shapes, max_iter=30, damping=e["damping"],
                          record=record)
        assert res["n_iter"] == e["n_iter"]
        tr = e["traj"]
        for it, (msg, rs, vs) in enumerate(recs[:len(tr["v:x"])]):
            for (s, t), (a, b) in msg.items():
                np.testing.assert_allclose(a, tr[f"{s}->{t}:a"][it], rtol=1e-9)
                np.testing.assert_allclose(b, tr[f"{s}->{t}:b"][it], rtol=1e-8, atol=1e-9)
            for v in shapes:
                np.testing.assert_allclose(rs[v], tr["r:" + v][it], rtol=1e-8, atol=1e-9)
                np.testing.assert_allclose(vs[v], tr["v:" + v][it], rtol=1e-9)


def test_gauss_channel_hidden_factor_oracle_matches_reference(golden_tree):
    "noisy 1-bit CS: GaussBernoulli -> Linear -> GaussianChannel -> SgnLikelihood (gaussian_channel.py:23-33)"
    for c in golden_tree["gch"]:
        N, M = c["N"], int(c["alpha"] * c["N"])
        W = np.random.RandomState(c["seed"]).randn(M, N) / np.sqrt(N)
        g = orc.GaussChannelFactor(c["var"])
        bz = c["msgs"]["bz"]
        for got, want in ((g.forward(1.3, bz, 0.7, 2 * bz), c["msgs"]["fwd"]), (g.backward(1.3, bz, 0.7, 2 * bz), c["msgs"]["bwd"])):
            close(got[0], want[0], rtol=1e-14)
            close(got[1], want[1], rtol=1e-14)
        fs = [orc.GBPriorFactor(N, rho=0.3), orc.LinearFactor(W), g, orc.SgnLikFactor(c["y"])]
        recs = []

        def record(it, fwd, bwd, rs_, vs_):
            recs.append(dict(r=[r.copy() for r in rs_], v=list(vs_), fwd=[(a, b.copy()) for a, b in fwd]))
        with np.errstate(all="ignore"):
            res = orc.ep_chain(fs, [(N,), (M,), (M,)], max_iter=40, damping=c["damping"], record=record)
        assert res["n_iter"] == c["n_iter"]
        t = c["traj"]
        for it in range(t["v:x"].shape[0]):
            for i, vid in enumerate(("x", "z", "a")):
                close(recs[it]["v"][i], t[f"v:{vid}"][it], rtol=1e-6)
                close(recs[it]["r"][i], t[f"r:{vid}"][it], rtol=1e-6, atol=1e-6 * np.abs(t[f"r:{vid}"][it]).max())
                close(recs[it]["fwd"][i][0], t[f"fwd:{vid}:a"][it], rtol=1e-6)
        close(res["r"][0], c["final_r"], rtol=1e-4, atol=1e-5)


def test_tree_se_oracle_matches_reference(golden_tree):
    "state evolution on the sparse-gradient tree: every scalar message and variance per iteration"
    for c in golden_tree["se"]:
        shape = tuple(c["shape"])
        d = len(shape)
        factors = dict(GaussianPrior=orc.GaussPriorFactor(shape), GaussianLikelihood=orc.GaussLikFactor(None, c["noise_var"]),
                       GradientChannel=orc.GradientFactor(shape),
                       GaussBernoulliPrior=orc.GBPriorFactor((d,) + shape, rho=c["rho"]))
        recs = []
        res = orc.se_tree(factors, ["x", "z"], TREE_EDGES, max_iter=12, damping=c["damping"], tol=0.0,
                          record=lambda it, msg, vs: recs.append((msg, vs)))
        for k, tau in c["tau"].items():
            close(res["tau"][k], tau, rtol=1e-12)
        t = c["traj"]
        for it, (msg, vs) in enumerate(recs):
            for (s, tt), a in msg.items():
                close(a, t[f"{s}->{tt}:a"][it], rtol=1e-7)
            for v in ("x", "z"):
                close(vs[v], t["v:" + v][it], rtol=1e-7)


def test_l21_prior_and_tv_oracle_match_reference(golden_tree):
    "MAP_L21NormPrior (axis 0) and total-variation denoising on the gradient tree"
    for c in golden_tree["l21"]:
        r, v = orc.l21_prior_posterior(c["ax"], c["bx"], c["gamma"])
        close(r, c["r"], rtol=1e-14, atol=1e-15)
        close(v, c["v"], rtol=1e-14)
        a, b = orc.l21_prior_message(c["ax"], c["bx"], c["gamma"])
        close(a, c["a_new"], rtol=1e-13)
        close(b, c["b_new"], rtol=1e-12, atol=1e-13)
        assert np.array_equal(orc.l21_support_mask(c["bx"], c["gamma"]), c["mask"])
    for e in golden_tree["tv"]:
        shape = tuple(e["shape"])
        d = len(shape)
        factors = dict(GaussianPrior=orc.GaussPriorFactor(shape), GaussianLikelihood=orc.GaussLikFactor(e["y"], e["noise_var"]),
                       GradientChannel=orc.GradientFactor(shape),
                       GaussBernoulliPrior=orc.L21PriorFactor((d,) + shape, gamma=e["gamma"]))
        edges = [(s.replace("GaussBernoulliPrior", "MAP_L21NormPrior"), t) for s, t in TREE_EDGES]
        factors["MAP_L21NormPrior"] = factors.pop("GaussBernoulliPrior")
        recs = []
        with np.errstate(all="ignore"):
            res = orc.ep_tree(factors, ["x", "z"], edges, dict(x=shape, z=(d,) + shape), max_iter=30,
                              damping=e["damping"], a0=e["a0"], b0=e["b0"],
                              record=lambda it, msg, rs, vs: recs.append((dict(msg), dict(rs), dict(vs))))
        assert res["n_iter"] == e["n_iter"]
        tr = e["traj"]
        for it in range(len(tr["v:x"])):
            msg, rs, vs = recs[it]
            for (s, t), (a, b) in msg.items():
                close(a, tr[f"{s}->{t}:a"][it], rtol=1e-8)
                close(b, tr[f"{s}->{t}:b"][it], rtol=1e-7, atol=1e-8 * max(1.0, np.abs(tr[f"{s}->{t}:b"][it]).max()))
            for v in ("x", "z"):
                close(rs[v], tr["r:" + v][it], rtol=1e-7, atol=1e-8)
                close(vs[v], tr["v:" + v][it], rtol=1e-8)


def vae_factors(c):
    D, H, N = c["dims"]
    return ([orc.GaussPriorFactor(D), orc.LinearFactor(c["W1"]), orc.BiasFactor(c["b1"]), orc.PWLFactor("l-relu", 0.1),
             orc.LinearFactor(c["W2"]), orc.BiasFactor(c["b2"]), orc.Pwl3Factor("h-tanh"),
             orc.GaussLikFactor(c["y"], c["noise_var"])],
            [(D,), (H,), (H,), (H,), (N,), (N,), (N,)], ["z", "a1", "a2", "a3", "a4", "a5", "x"])


def test_generative_prior_chain_oracle_matches_reference(golden_tree):
    "VAE-like prior (examples/vae_prior/plot_vae.py): Linear -> Bias -> LeakyRelu -> Linear -> Bias -> HardTanh"
    for c in golden_tree["vae"]:
        g = orc.BiasFactor(c["b1"])
        bb = c["msgs"]["b"]
        for got, want in ((g.forward(1.3, bb, 0.7, 2 * bb), c["msgs"]["fwd"]), (g.backward(1.3, bb, 0.7, 2 * bb), c["msgs"]["bwd"])):
            close(got[0], want[0], rtol=1e-14)
            close(got[1], want[1], rtol=1e-14)
        close(g.second_moment(1.7), c["msgs"]["tau"], rtol=1e-14)
        fs, shapes, ids = vae_factors(c)
        recs = []
        with np.errstate(all="ignore"):
            orc.ep_chain(fs, shapes, max_iter=10, damping=c["damping"], tol=0.0, max_increase=np.inf,
                         record=lambda it, fwd, bwd, rs_, vs_: recs.append(([r.copy() for r in rs_], list(vs_))))
        t = c["traj"]
        for it in range(t["v:x"].shape[0]):
            for i, vid in enumerate(ids):
                close(recs[it][1][i], t[f"v:{vid}"][it], rtol=1e-6)
                close(recs[it][0][i], t[f"r:{vid}"][it], rtol=1e-6, atol=1e-6 * max(1.0, np.abs(t[f"r:{vid}"][it]).max()))


def committee_graph(c):
    "names as the generator's: variables by id, factors by class and surrounding variable ids"
    K, N, M = c["K"], c["N"], c["M"]
    F = np.random.RandomState(17 + K).standard_normal((M, N)) / np.sqrt(N)
    factors, edges, shapes = {}, [], {"n": (M,)}
    summed = ",".join(sorted([f"a_{k}" for k in range(K)] + ["n"]))
    for k in range(K):
        p, lin, act = f"GaussianPrior[x_{k}]", f"LinearChannel[x_{k},z_{k}]", f"ReluChannel[a_{k},z_{k}]"
        factors[p] = orc.GaussPriorFactor(N, var=c["pvars"][k])
        factors[lin] = orc.LinearFactor(F)
        factors[act] = orc.PWLFactor("relu")
        edges += [(p, f"x_{k}"), (f"x_{k}", lin), (lin, f"z_{k}"), (f"z_{k}", act), (act, f"a_{k}"),
                  (f"a_{k}", f"SumChannel[{summed}]")]
        shapes.update({f"x_{k}": (N,), f"z_{k}": (M,), f"a_{k}": (M,)})
    factors[f"SumChannel[{summed}]"] = orc.SumFactor(K)
    factors["GaussianLikelihood[n]"] = orc.GaussLikFactor(c["y"], c["noise_var"])
    edges += [(f"SumChannel[{summed}]", "n"), ("n", "GaussianLikelihood[n]")]
    return F, factors, edges, shapes


def test_committee_machine_oracle_matches_reference(golden_tree):
    "K experts summed by the K-input SumChannel (committee_model.py:8-41): every message, every iteration"
    for c in golden_tree["com"]:
        m = c["msgs"]
        s = orc.SumFactor(c["K"])
        fa, fb = s.forward(m["az"], m["bz"], m["ax"], m["bx"])
        close(fa, m["fwd"][0], rtol=1e-14)
        close(fb, m["fwd"][1], rtol=1e-13)
        for (a, b), wa, wb in zip(s.backward(m["az"], m["bz"], m["ax"], m["bx"]), m["bwd"][0], m["bwd"][1]):
            close(a, wa, rtol=1e-14)
            close(b, wb, rtol=1e-13)
        F, factors, edges, shapes = committee_graph(c)
        recs = []
        orc.ep_tree(factors, list(shapes), edges, shapes, max_iter=12, damping=c["damping"], tol=0.0, max_increase=np.inf,
                    record=lambda it, msg, rs, vs: recs.append((dict(msg), dict(rs), dict(vs))))
        tr = c["traj"]
        for it, (msg, rs, vs) in enumerate(recs):
            tol = 1e-8 if it < 6 else 1e-6
            for (s_, t), (a, b) in msg.items():
                close(a, tr[f"{s_}->{t}:a"][it], rtol=tol)
                close(b, tr[f"{s_}->{t}:b"][it], rtol=tol, atol=tol * max(1.0, np.abs(tr[f"{s_}->{t}:b"][it]).max()))
            for v in shapes:
                close(rs[v], tr["r:" + v][it], rtol=tol, atol=tol)
                close(vs[v], tr["v:" + v][it], rtol=tol)


def sgn_committee_graph(c):
    K, N, M = c["K"], c["N"], c["M"]
    F = np.random.RandomState(29).standard_normal((M, N)) / np.sqrt(N)
    factors, edges, shapes = {}, [], {"a": (M,)}
    summed = ",".join(sorted([f"a_{k}" for k in range(K)] + ["a"]))
    for k in range(K):
        p, lin, act = f"BinaryPrior[x_{k}]", f"LinearChannel[x_{k},z_{k}]", f"SgnChannel[a_{k},z_{k}]"
        factors[p] = orc.BinaryPriorFactor(N, p_pos=c["p_pos"][k])
        factors[lin] = orc.LinearFactor(F)
        factors[act] = orc.PWLFactor("sgn")
        edges += [(p, f"x_{k}"), (f"x_{k}", lin), (lin, f"z_{k}"), (f"z_{k}", act), (act, f"a_{k}"),
                  (f"a_{k}", f"SumChannel[{summed}]")]
        shapes.update({f"x_{k}": (N,), f"z_{k}": (M,), f"a_{k}": (M,)})
    factors[f"SumChannel[{summed}]"] = orc.SumFactor(K)
    factors["SgnLikelihood[a]"] = orc.SgnLikFactor(c["y"])
    edges += [(f"SumChannel[{summed}]", "a"), ("a", "SgnLikelihood[a]")]
    return F, factors, edges, shapes


def test_binary_prior_and_sgn_committee_oracle_match_reference(golden_tree):
    for c in golden_tree["binp"]:
        r, v = orc.binary_prior_posterior(1.3, c["bx"], c["p_pos"])
        close(r, c["r"], rtol=1e-14)
        close(v, c["v"], rtol=1e-14)
        a, b = orc.BinaryPriorFactor(9, c["p_pos"]).forward(None, None, 1.3, c["bx"])
        close(a, c["a_new"], rtol=1e-13)
        close(b, c["b_new"], rtol=1e-12, atol=1e-13)
    c = golden_tree["sgncom"][0]
    F, factors, edges, shapes = sgn_committee_graph(c)
    recs = []
    with np.errstate(all="ignore"):
        orc.ep_tree(factors, list(shapes), edges, shapes, max_iter=10, damping=c["damping"], tol=0.0, max_increase=np.inf,
                    record=lambda it, msg, rs, vs: recs.append((dict(msg), dict(rs), dict(vs))))
    tr = c["traj"]
    for it, (msg, rs, vs) in enumerate(recs):
        tol = 1e-7 if it < 5 else 1e-5
        for (s_, t), (a, b) in msg.items():
            close(a, tr[f"{s_}->{t}:a"][it], rtol=tol)
            close(b, tr[f"{s_}->{t}:b"][it], rtol=tol, atol=tol * max(1.0, np.abs(tr[f"{s_}->{t}:b"][it]).max()))
        for v in shapes:
            close(rs[v], tr["r:" + v][it], rtol=tol, atol=tol)


def test_committee_se_oracle_matches_reference(golden_tree):
    "state evolution through the K-input SumChannel (soft committee, Marchenko-Pastur linear layers)"
    for c in golden_tree["comse"]:
        K = c["K"]
        factors, edges, variables = {}, [], ["n"]
        summed = ",".join(sorted([f"a_{k}" for k in range(K)] + ["n"]))
        for k in range(K):
            p, lin, act = f"GaussianPrior[x_{k}]", f"MarchenkoPasturChannel[x_{k},z_{k}]", f"ReluChannel[a_{k},z_{k}]"
            factors[p] = orc.GaussPriorFactor(None, var=c["pvars"][k])
            factors[lin] = orc.MPFactor(c["alpha"])
            factors[act] = orc.PWLFactor("relu")
            edges += [(p, f"x_{k}"), (f"x_{k}", lin), (lin, f"z_{k}"), (f"z_{k}", act), (act, f"a_{k}"),
                      (f"a_{k}", f"SumChannel[{summed}]")]
            variables += [f"x_{k}", f"z_{k}", f"a_{k}"]
        factors[f"SumChannel[{summed}]"] = orc.SumFactor(K)
        factors["GaussianLikelihood[n]"] = orc.GaussLikFactor(None, c["noise_var"])
        edges += [(f"SumChannel[{summed}]", "n"), ("n", "GaussianLikelihood[n]")]
        recs = []
        res = orc.se_tree(factors, variables, edges, max_iter=3, damping=c["damping"], tol=0.0, method="gl",
                          record=lambda it, msg, vs: recs.append((msg, vs)))
        for k, tau in c["tau"].items():
            close(res["tau"][k], tau, rtol=1e-12)
        t = c["traj"]
        for it, (msg, vs) in enumerate(recs):
            for (s_, tt), a in msg.items():
                close(a, t[f"{s_}->{tt}:a"][it], rtol=2e-5)
            for v in variables:
                close(vs[v], t["v:" + v][it], rtol=2e-5)


def test_binary_prior_se_oracle_matches_reference(golden_tree):
    for c in golden_tree["binse"]:
        close(orc.se_binary_prior_error(c["ax"], c["p_pos"]), c["v"], rtol=1e-8)
        close(orc.se_binary_prior_error(c["ax"], c["p_pos"], method="gl"), c["v"], rtol=1e-6)


def test_linear_solve_oracle_matches_reference_goldens():
    """LinearChannel(precompute_svd=False): the oracle's restatement against the reference's own outputs
    (tests/golden/solve.npz from tools/make_golden_solve.py), including the ascending-spectrum `singular`."""
    from tests.conftest import load_golden
    from oracle import tramp_oracle as orc
    cases = load_golden("solve")
    assert len(cases) == 20
    for c in cases:
        o = orc.LinearSolveOracle(c["W"])
        assert o.rank == c["rank"]
        np.testing.assert_allclose(o.spectrum, c["spectrum"], rtol=0, atol=1e-13 * max(1.0, c["spectrum"].max()))
        az, ax, bz, bx = c["az"], c["ax"], c["bz"], c["bx"]
        np.testing.assert_allclose(o.n_eff(az, ax), c["n_eff"], rtol=1e-9)
        np.testing.assert_allclose(o.backward_mean(az, bz, ax, bx), c["rz"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o.forward_mean(az, bz, ax, bx), c["rx"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o.backward_variance(az, ax), c["vz"], rtol=1e-9)
        np.testing.assert_allclose(o.forward_variance(az, ax), c["vx"], rtol=1e-9)
        a_b, b_b = o.backward_message(az, bz, ax, bx)
        a_f, b_f = o.forward_message(az, bz, ax, bx)
        np.testing.assert_allclose([a_b, a_f], [c["a_bwd"], c["a_fwd"]], rtol=1e-9)
        np.testing.assert_allclose(b_b, c["b_bwd"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(b_f, c["b_fwd"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(o.log_partition(az, bz, ax, bx), c["logZ"], rtol=1e-10)
    # the quirk is real: for the wide matrix the variances differ from the precompute_svd=True channel
    wide = next(c for c in cases if c["name"] == "wide_8x12")
    assert abs(orc.LinearOracle(wide["W"]).backward_variance(wide["az"], wide["ax"]) - wide["vz"]) > 1e-3


def test_complex_conv_and_dft_oracles_match_reference_goldens():
    """real=False: ConvChannel and DFTChannel on planar complex vectors, the oracle's restatement against the reference's
    own outputs (tests/golden/conv_complex.npz from tools/make_golden_conv_complex.py)."""
    import os
    from tests.conftest import GOLDEN
    from oracle import tramp_oracle as orc
    with np.load(os.path.join(GOLDEN, "conv_complex.npz"), allow_pickle=True) as f:
        conv, dft = list(f["data"]), list(f["dft"])
    assert len(conv) == 16 and len(dft) == 8
    for c in conv:
        o = orc.ComplexConvOracle(c["filter"])
        az, ax, bz, bx = c["az"], c["ax"], c["bz"], c["bx"]
        np.testing.assert_allclose(o.n_eff(az, ax), c["n_eff"], rtol=1e-12)
        np.testing.assert_allclose(o.second_moment(1.7), c["tau"], rtol=1e-12)
        np.testing.assert_allclose(o.backward_mean(az, bz, ax, bx), c["rz"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(o.forward_mean(az, bz, ax, bx), c["rx"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose([o.backward_variance(az, ax), o.forward_variance(az, ax)], [c["vz"], c["vx"]], rtol=1e-12)
        a_b, b_b = o.backward_message(az, bz, ax, bx)
        a_f, b_f = o.forward_message(az, bz, ax, bx)
        np.testing.assert_allclose([a_b, a_f], [c["a_bwd"], c["a_fwd"]], rtol=1e-10)
        np.testing.assert_allclose(b_b, c["b_bwd"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(b_f, c["b_fwd"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(o.convolve(bz), c["sample"], rtol=1e-10, atol=1e-12)
    for c in dft:
        o = orc.ComplexDFTOracle(len(c["shape"]))
        a_f, b_f = o.forward_message(1.3, c["bz"], 0.7, c["bx"])
        a_b, b_b = o.backward_message(1.3, c["bz"], 0.7, c["bx"])
        assert (a_f, a_b) == (c["a_fwd"], c["a_bwd"])
        np.testing.assert_allclose(b_f, c["b_fwd"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(b_b, c["b_bwd"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(b_f, c["sample"], rtol=1e-12, atol=1e-13)


def test_adaptive_damping_oracle_matches_reference_trajectories():
    """ep_chain_adaptive (the four-messages-per-variable step search of message_passing.py:151-185) against the
    reference's own trajectories (tests/golden/adaptive.npz, tools/make_golden_adaptive.py): every message of every
    edge and the posteriors, iteration by iteration."""
    from tests.conftest import load_golden
    from oracle import tramp_oracle as orc
    cases = load_golden("adaptive")
    assert len(cases) == 4
    for c in cases:
        N, M = c["N"], int(c["alpha"] * c["N"])
        rs = np.random.RandomState(c["seed"])
        W = rs.randn(M, N) / np.sqrt(N)
        assert W.sum() == c["W_sum"]
        lik = {"gaussian": lambda: orc.GaussLikFactor(c["y"], 0.01), "sgn": lambda: orc.SgnLikFactor(c["y"]),
               "abs": lambda: orc.AbsLikFactor(c["y"])}[c["output"]]()
        recs = []
        orc.ep_chain_adaptive([orc.GBPriorFactor(N, rho=c["rho"]), orc.LinearFactor(W), lik], [N, M],
                              max_iter=c["n_iter"], record=lambda it, st: recs.append(st))
        assert len(recs) == c["n_iter"]
        damped = sum(1 for b in c["betas"] if b[3] < 1 and b[4])
        assert damped >= 3                                       # the goldens do exercise the search
        for it, st in enumerate(recs):
            t = c["traj"][it]
            rt = 1e-9 if it < 3 else 1e-7
            for i, vid in enumerate(("x", "z")):
                for key, m in ((f"fwd:f->{vid}", st["ff"][i]), (f"fwd:{vid}->f", st["vf"][i]),
                               (f"bwd:f->{vid}", st["fb"][i]), (f"bwd:{vid}->f", st["vb"][i])):
                    np.testing.assert_allclose(m[0], t[key][0], rtol=rt, err_msg=f"{c['output']} it {it} {key} a")
                    np.testing.assert_allclose(m[1], t[key][1], rtol=rt, atol=rt * np.abs(t[key][1]).max(),
                                               err_msg=f"{c['output']} it {it} {key} b")
                np.testing.assert_allclose(st["r"][i], t[f"r:{vid}"], rtol=rt, atol=rt * np.abs(t[f"r:{vid}"]).max())
                np.testing.assert_allclose(st["v"][i], t[f"v:{vid}"], rtol=rt)


# -- the rest of the piecewise-linear family: log-partitions, and the AsymmetricAbs likelihood (section 8f rank 2)
def test_f2_log_partitions_oracle_matches_reference(golden_f2_logz):
    for c in golden_f2_logz:
        if c["family"] == "channel":
            reg = orc.f2_regions(c["name"], **c["kwargs"])
            close(orc.region_channel_log_partition(reg, c["az"], c["bz"], c["ax"], c["bx"]), c["A"], rtol=1e-11)
        else:
            reg = orc.f2_regions(c["name"], likelihood=True, **c["kwargs"])
            A = orc.region_lik_log_partition(reg, c["az"], c["bz"], c["y"])
            r, v = orc.region_lik_posterior(reg, c["az"], c["bz"], c["y"])
            if c["name"] == "a-abs":
                # the reference as written: np.log(-1) on the descending branch poisons every output
                assert np.isnan(c["A"]) and np.isnan(A) and np.isnan(c["v"]) and np.isnan(v) and np.isnan(c["r"]).all()
            else:
                close(A, c["A"], rtol=1e-11)
                close(r, c["r"], rtol=1e-10, atol=1e-12); close(v, c["v"], rtol=1e-10)


def test_aabs_likelihood_with_the_jacobian_is_the_abs_likelihood_away_from_the_kink():
    """with log|slope| the two branches of AsymmetricAbs weigh as softmax(+-bz y): for y > |shift| the posterior is
    AbsLikelihood's (abs_likelihood.py:31-36); -shift < y <= shift pins z = -y; y <= -shift is incompatible (NaN)"""
    rs = np.random.RandomState(1)
    for shift in (1e-4, 0.3):
        reg = orc.f2_regions("a-abs", shift=shift)
        y = np.abs(rs.standard_normal(200)) + shift + 1e-3
        bz = 2 * rs.standard_normal(200)
        r, v = orc.region_lik_posterior(reg, 1.3, bz, y, isotropic=False, abs_slope=True)
        r_abs, v_abs = orc.abs_lik_posterior(1.3, bz, y, isotropic=False)
        close(r, r_abs, rtol=1e-12); close(v, v_abs, rtol=1e-9, atol=1e-12)      # y^2 (1 - tanh^2) cancels where tanh -> 1
        A = orc.region_lik_log_partition(reg, 1.3, bz, y, abs_slope=True)
        close(A, orc.abs_lik_log_partition(1.3, bz, y), rtol=1e-12)
        y2 = np.array([shift, 0.5 * shift, 0.0, -0.5 * shift])
        r2, v2 = orc.region_lik_posterior(reg, 1.3, np.ones(4), y2, isotropic=False, abs_slope=True)
        close(r2, -y2); close(v2, 0 * y2)
        r3, _ = orc.region_lik_posterior(reg, 1.3, np.ones(2), np.array([-shift, -1.0]), isotropic=False, abs_slope=True)
        assert np.isnan(r3).all()
    # SE error: the abs likelihood's minus the slab |z| <= shift
    close(orc.se_aabs_lik_error(1.7, 1.0, shift=1e-4), orc.se_abs_lik_error(1.7, 1.0), rtol=1e-5)
