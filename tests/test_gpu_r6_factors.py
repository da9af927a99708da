"""GPU parity of the factors added in round 6, through the C ABI, against the reference's goldens
(tests/golden/priors2.npz, complex.npz: tools/make_golden_r6.py) and the oracle:
ExponentialPrior / PositivePrior / GaussianMixturePrior (tramp/priors/{exponential,positive,gaussian_mixture}_prior.py),
ModulusLikelihood (tramp/likelihoods/modulus_likelihood.py), ComplexLinearChannel
(tramp/channels/linear/complex_linear_channel.py) and the phase-retrieval EP they make."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc
from tests.conftest import prior2_oracle, redraw_prior2_W, redraw_complex_W
from tests.test_gpu_ep import Recorder

pytestmark = pytest.mark.gpu

RT = {np.float64: 1e-9, np.float32: 2e-4}       # north_star: 1e-6 relative f64 / 1e-3 f32 on posterior means


def close(a, b, rtol, scale=None):
    b = np.asarray(b, dtype=float)
    sc = np.nanmax(np.abs(b)) if scale is None else scale
    np.testing.assert_allclose(np.asarray(a, dtype=float), b, rtol=rtol, atol=rtol * max(sc, 1e-300), equal_nan=True)


def make_prior(T, ctx, name, kw, size, dtype=np.float64, isotropic=True):
    cls = dict(exponential=T.ExponentialPrior, positive=T.PositivePrior, mixture=T.GaussianMixturePrior)[name]
    return cls(size=size, isotropic=isotropic, dtype=dtype, ctx=ctx, **kw)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_prior2_against_reference_goldens(ctx, golden_priors2, dtype):
    import tramp_amd as T
    rt, n_run = RT[dtype], 0
    for c in golden_priors2["cases"]:
        if c.get("diag"):
            continue
        bx, ref = c["bx"], c
        if dtype == np.float32:
            # f32: without the eight edge inputs (|b| up to 300: x = |b| / sqrt(2 a) in the hundreds, where the reference's
            # v0 (1 + g2 - g1^2) needs eps x^2 << 1) and inside f32's range; the expected values from the oracle, which the
            # CPU suite pins on these same goldens
            if np.abs(bx[8:]).max() / min(1.0, c["ax"]) > 1e4 or c["ax"] > 1e3:
                continue
            bx = bx[8:].astype(np.float32)
            post, _, _ = prior2_oracle(orc, c["name"], c["kwargs"])
            r_o, v_o = post(c["ax"], bx.astype(np.float64))
            a_o, b_o = orc.ab_new(r_o, v_o, c["ax"], bx.astype(np.float64))
            ref = dict(r=r_o, v=v_o, a_new=a_o, b_new=b_o)
        pr = make_prior(T, ctx, c["name"], c["kwargs"], bx.size, dtype)
        tag = (c["name"], c["kwargs"], c["ax"])
        r, v = pr.compute_forward_posterior(c["ax"], bx)
        close(r, ref["r"], rt)
        # far tail (b = -300 at a = 1e-3): the reference's own variance formula cancels to eps x^2 - 3e-9 of the mean there
        close(v, ref["v"], max(rt, 1e-8))
        a_new, b_new = pr.compute_forward_message(c["ax"], bx)
        close(a_new, ref["a_new"], max(rt, 1e-7), scale=ref["a_new"] + c["ax"])
        # (f32: r (a + a_new) - b cancels, compared on the scale of its terms)
        close(b_new, ref["b_new"], max(rt, 1e-7), scale=None if dtype == np.float64 else max(np.abs(ref["b_new"]).max(), np.abs(bx).max()))
        if dtype == np.float64:
            assert abs(pr.compute_log_partition(c["ax"], bx) - c["A"]) <= 1e-10 * max(1.0, abs(c["A"])), tag
            assert pr.second_moment() == pytest.approx(c["second_moment"], rel=1e-14)
        n_run += 1
    assert n_run == (35 if dtype == np.float64 else 28), n_run


def test_prior2_diagonal_beliefs(ctx, golden_priors2):
    "isotropic=False (the mode of the reference's own prior tests, tests/test_priors.py:17-51)"
    import tramp_amd as T
    n_run = 0
    for c in golden_priors2["cases"]:
        if not c.get("diag"):
            continue
        pr = make_prior(T, ctx, c["name"], c["kwargs"], c["bx"].size, isotropic=False)
        r, v = pr.compute_forward_posterior(c["ax"], c["bx"])
        close(r, c["r"], 1e-11)
        close(v, c["v"], 1e-9)
        a_new, b_new = pr.compute_forward_message(c["ax"], c["bx"])
        close(a_new, c["a_new"], 1e-8)
        close(b_new, c["b_new"], 1e-8)
        post, _, logz = prior2_oracle(orc, c["name"], c["kwargs"])
        # the reference's diagonal log-partition is the mean of per-component terms: the oracle's isotropic form per component
        A = np.mean([logz(a, np.array([b])) for a, b in zip(c["ax"][:40], c["bx"][:40])])
        assert abs(pr.compute_log_partition(c["ax"][:40], c["bx"][:40]) - A) <= 1e-10 * max(1.0, abs(A))
        n_run += 1
    assert n_run == 7


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,kw", [("exponential", dict(mean=0.7)), ("positive", {}),
                                     ("mixture", dict(probs=[0.2, 0.5, 0.3], means=[-2.0, 0.0, 1.5], vars=[0.3, 0.01, 2.0])),
                                     ("mixture", dict(probs=[0.125] * 8, means=list(np.linspace(-3, 3, 8)), vars=[0.2] * 8))])
@pytest.mark.parametrize("n,batch", [(1000, 1), (4096, 3), (100003, 2), (1 << 20, 2), (300000, 5)])
def test_prior2_every_kernel_form_against_oracle(ctx, name, kw, n, batch, dtype):
    """ragged and vector lengths, one and several instances with their own precisions: the single-workgroup, tile, one-launch
    and grid-stride forms of the template all see the new ops; damping fused"""
    import tramp_amd as T
    rs = np.random.RandomState(n % 977 + batch)
    ax = np.exp(rs.uniform(np.log(0.3), np.log(8.0), batch))
    bx = (ax[:, None] * rs.standard_normal((batch, n)) * 1.2 + 0.3).astype(dtype)
    pr = make_prior(T, ctx, name, kw, n, dtype)
    post, _, _ = prior2_oracle(orc, name, kw)
    rt = RT[dtype]
    a_new, b_new = pr.compute_forward_message(ax, bx)
    r, v = pr.compute_forward_posterior(ax, bx)
    a_old, b_old = np.full(batch, 0.4), 0.1 * rs.standard_normal((batch, n)).astype(dtype)
    a_d, b_d = pr.compute_forward_message(ctx.array(ax), ctx.array(bx, dtype), damping=0.3, old_a=ctx.array(a_old),
                                          old_b=ctx.array(b_old, dtype))
    for b in range(batch):
        ro, vo = post(ax[b], bx[b].astype(np.float64))
        close(r[b], ro, rt)
        close(v[b], vo, max(rt, 1e-9))
        an, bn = orc.ab_new(ro, vo, ax[b], bx[b].astype(np.float64))
        close(a_new[b], an, max(rt, 1e-7), scale=an + ax[b])
        close(b_new[b], bn, max(rt, 1e-7))
        close(a_d.get()[b], 0.3 * a_old[b] + 0.7 * an, max(rt, 1e-7), scale=an + ax[b])
        close(b_d.get()[b], 0.3 * b_old[b].astype(np.float64) + 0.7 * bn, max(rt, 1e-7))


def test_mixture_se_error(ctx, golden_priors2):
    import tramp_amd as T
    for c in golden_priors2["se"]:
        pr = T.GaussianMixturePrior(size=None, ctx=ctx, **c["kwargs"])
        assert pr.compute_forward_error(c["ax"]) == pytest.approx(c["v"], rel=1e-7)
        assert pr.compute_forward_state_evolution(c["ax"]) == pytest.approx(c["a_new"], rel=1e-6)
    pr = T.GaussianMixturePrior(size=None, ctx=ctx, **golden_priors2["se"][0]["kwargs"])
    axs = np.array([c["ax"] for c in golden_priors2["se"][:5]])
    close(pr.compute_forward_error(axs), [c["v"] for c in golden_priors2["se"][:5]], 1e-7)


def test_error_conventions_of_the_new_priors(ctx):
    import tramp_amd as T
    with pytest.raises(AssertionError):
        T.ExponentialPrior(size=4, mean=-1.0, ctx=ctx)                   # exponential_prior.py:21
    with pytest.raises(NotImplementedError):
        T.ExponentialPrior(size=4, ctx=ctx).compute_forward_error(1.0)    # :73-74
    with pytest.raises(NotImplementedError):
        T.PositivePrior(size=4, ctx=ctx).compute_forward_error(1.0)
    with pytest.raises(NotImplementedError):
        T.GaussianMixturePrior(size=4, probs=[0.1] * 10, means=[0] * 10, vars=[1] * 10, ctx=ctx)
    with pytest.raises(T._lib.TrampHipError):
        T.GaussianMixturePrior(size=4, probs=[0.5, 0.5], means=[0, 1], vars=[1, -1.0], ctx=ctx).compute_forward_message(
            1.0, np.zeros(4))
    assert isinstance(T.get_prior(5, "positive", ctx=ctx), T.ExponentialPrior)      # the reference's registry, as found
    assert isinstance(T.get_prior(5, "mixture", ctx=ctx), T.GaussianMixturePrior)
    # NaN in, NaN out (and a_new NaN, which the EP monitor turns into the reference's ValueError)
    a, b = T.PositivePrior(size=8, ctx=ctx).compute_forward_message(1.0, np.array([0, 1, np.nan, 2, 3, 4, 5, 6.0]))
    assert np.isnan(a) and np.isnan(b[2])


def run_traj(T, ctx, model, c, ids, init, max_iter):
    ep = T.ExpectationPropagation(model, ctx=ctx)
    rec = Recorder(T.EarlyStoppingEP())
    ep.iterate(max_iter=max_iter, callback=rec, damping=c["damping"], initializer=init)
    assert ep.n_iter == c["n_iter"], (c.get("name"), c["N"], c["damping"])
    t = c["traj"]
    n = min(t[f"v:{ids[0]}"].shape[0], len(rec.iters))
    for it in range(n):
        rt = 1e-7 if it < 6 else 1e-5
        for vid in ids:
            e = rec.iters[it]["edges"]
            a_tot = t[f"fwd:{vid}:a"][it] + t[f"bwd:{vid}:a"][it]
            close(e[(vid, "fwd")][0], t[f"fwd:{vid}:a"][it], rt, scale=a_tot)
            close(e[(vid, "bwd")][0], t[f"bwd:{vid}:a"][it], rt, scale=a_tot)
            for d in ("fwd", "bwd"):
                ref = t[f"{d}:{vid}:b"][it]
                # (a message whose precision is the floor 1e-11 is round-off of O(1) terms: an absolute floor on the scale)
                close(np.asarray(rec.iters[it]["b"][(vid, d)]).reshape(ref.shape), ref, rt, scale=max(np.abs(ref).max(), 1e-7))
            ref = t[f"r:{vid}"][it]
            close(np.asarray(rec.iters[it]["vars"][vid]["r"]).reshape(ref.shape), ref, rt, scale=max(np.abs(ref).max(), 1e-7))
            close(rec.iters[it]["vars"][vid]["v"], t[f"v:{vid}"][it], rt)
    return ep


def test_prior2_ep_trajectories(ctx, golden_priors2):
    "every message of the reference's iterate() behind a dense LinearChannel, with and without damping"
    import tramp_amd as T
    assert len(golden_priors2["ep"]) == 14
    for c in golden_priors2["ep"]:
        W = redraw_prior2_W(c)
        model = (make_prior(T, ctx, c["name"], c["kwargs"], c["N"]) @ T.SISOVariable(id="x")
                 @ T.LinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z")
                 @ T.GaussianLikelihood(y=c["y"], var=c["noise_var"], ctx=ctx)).to_model()
        run_traj(T, ctx, model, c, ["x", "z"], T.CustomInit(a=c["a_init"], b=0), 30)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_modulus_likelihood_against_reference_goldens(ctx, golden_complex, dtype):
    import tramp_amd as T
    rt = RT[dtype]
    for c in golden_complex["modulus"]:
        if dtype == np.float32 and np.abs(c["bz"]).max() > 1e4:
            continue
        lk = T.ModulusLikelihood(y=c["y"], dtype=dtype, ctx=ctx)
        r, v = lk.compute_backward_posterior(c["az"], c["bz"], c["y"])
        assert r.shape == (2, c["y"].size)
        close(r, c["r"], rt)
        close(v, c["v"], rt)
        a_new, b_new = lk.compute_backward_message(c["az"], c["bz"])
        close(a_new, c["a_new"], max(rt, 1e-8), scale=c["a_new"] + c["az"])
        close(b_new, c["b_new"], max(rt, 1e-8))
        if dtype == np.float64:
            # the reference's own value is NaN on these inputs: scipy's ive(0, x) gives up beyond x ~ 1e9 (component 2:
            # |b| y = 1.5e9) and y = 0 (component 1) is log(0); without the two the oracle (pinned on the goldens, NaN
            # and all) is the check - and the device value WITH them is the finite / -inf one
            keep = np.r_[0, 3:c["y"].size]
            A = T.ModulusLikelihood(y=c["y"][keep], ctx=ctx).compute_log_partition(c["az"], c["bz"][:, keep])
            assert A == pytest.approx(orc.modulus_lik_log_partition(c["az"], c["bz"][:, keep], c["y"][keep]), rel=1e-11)
            assert np.isneginf(lk.compute_log_partition(c["az"], c["bz"])) and np.isnan(c["A"])
    # an ensemble: instance b == the single call on (az_b, bz_b, y_b); damping fused
    rs = np.random.RandomState(5)
    B, n = 3, 5000
    y = np.abs(rs.standard_normal((B, n)) + 1j * rs.standard_normal((B, n)))
    bz = 2.0 * rs.standard_normal((B, 2, n))
    az = np.array([0.7, 1.9, 6.0])
    lk = T.ModulusLikelihood(y=y, dtype=dtype, ctx=ctx)
    a_new, b_new = lk.compute_backward_message(az, bz)
    assert b_new.shape == (B, 2, n)
    for b in range(B):
        an, bn = orc.modulus_lik_message(az[b], bz[b].astype(dtype).astype(np.float64), y[b].astype(dtype).astype(np.float64))
        close(a_new[b], an, max(rt, 1e-8), scale=an + az[b])
        close(b_new[b], bn, max(rt, 1e-8))


def test_complex_linear_channel_against_reference_goldens(ctx, golden_complex):
    import tramp_amd as T
    for c in golden_complex["linear"]:
        ch = T.ComplexLinearChannel(c["W"], ctx=ctx, svd="host")
        assert (ch.Nx, ch.Nz, ch.rank) == (c["M"], c["N"], c["rank"])
        assert ch.compute_n_eff(c["az"], c["ax"]) == pytest.approx(c["n_eff"], rel=1e-10)
        rz, vz = ch.compute_backward_posterior(c["az"], c["bz"], c["ax"], c["bx"])
        rx, vx = ch.compute_forward_posterior(c["az"], c["bz"], c["ax"], c["bx"])
        assert rz.shape == (2, c["N"]) and rx.shape == (2, c["M"])
        close(rz, c["rz"], 1e-9)
        close(rx, c["rx"], 1e-9)
        assert vz == pytest.approx(c["vz"], rel=1e-10) and vx == pytest.approx(c["vx"], rel=1e-10)
        # b_new = r (a + a_new) - b cancels down to a_new / a of its terms (ax = 0: a_new is the floor 1e-11 and the
        # reference's own b_new is round-off of bz): compared on the scale of the terms
        a, b = ch.compute_backward_message(c["az"], c["bz"], c["ax"], c["bx"])
        close(a, c["a_bwd"], 1e-8, scale=c["a_bwd"] + c["az"])
        close(b, c["b_bwd"], 1e-8, scale=max(np.abs(c["b_bwd"]).max(), np.abs(c["bz"]).max() * 1e-4))
        a, b = ch.compute_forward_message(c["az"], c["bz"], c["ax"], c["bx"])
        close(a, c["a_fwd"], 1e-8, scale=c["a_fwd"] + c["ax"])
        close(b, c["b_fwd"], 1e-8, scale=max(np.abs(c["b_fwd"]).max(), np.abs(c["bx"]).max() * 1e-4))
        assert ch.compute_log_partition(c["az"], c["bz"], c["ax"], c["bx"]) == pytest.approx(c["A"], rel=1e-9)
        sp = np.sort(np.linalg.svd(c["W"], compute_uv=False)**2)[::-1]
        close(ch.spectrum[:sp.size], sp, 1e-10)


def test_phase_retrieval_ep_trajectories(ctx, golden_complex):
    """GaussBernoulliPrior(size=(2, N)) -> ComplexLinearChannel -> ModulusLikelihood: every message of the reference's
    iterate() from a symmetry-breaking initial message (CustomInit), with and without damping"""
    import tramp_amd as T
    for c in golden_complex["ep"]:
        W = redraw_complex_W(c)
        N = c["N"]
        model = (T.GaussBernoulliPrior(size=(2, N), rho=c["rho"], ctx=ctx) @ T.SISOVariable(id="x")
                 @ T.ComplexLinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z")
                 @ T.ModulusLikelihood(y=c["y"], ctx=ctx)).to_model()
        init = T.CustomInit(a_init=[("x", "bwd", c["a0"])], b_init=[("x", "bwd", c["b0"])], a=0, b=0)
        ep = run_traj(T, ctx, model, c, ["x", "z"], init, 25)
        assert ep.shapes == {"x": (2, N), "z": (2, c["M"])}


def test_phase_retrieval_model_api(ctx):
    """glm_generative(output_type='modulus') -> sample -> to_observed -> EP, as the reference's phase-retrieval examples
    (examples/glm/plot_complex_pr.py): the signal is recovered up to its global phase, in as many iterations as the oracle
    takes from the same symmetry-breaking initial message"""
    import tramp_amd as T
    np.random.seed(1)
    N = 64
    model = T.glm_generative(N=N, alpha=3.0, ensemble_type="complex_gaussian", prior_type="gauss_bernoulli",
                             output_type="modulus", prior_rho=0.5, ctx=ctx)
    sample = model.sample(seed=4)
    assert sample["x"].shape == (2, N) and sample["z"].shape == (2, 192) and sample["y"].shape == (192,)
    student = model.to_observed({"y": sample["y"]})
    assert [type(f).__name__ for f in student.factors] == ["GaussBernoulliPrior", "ComplexLinearChannel", "ModulusLikelihood"]
    b0 = np.random.RandomState(2).standard_normal((2, N))
    ep = T.ExpectationPropagation(student, ctx=ctx)
    # EarlyStoppingEP's give-up rule (an increase after iteration 5) fires on the non-monotone start of phase retrieval
    ep.iterate(max_iter=300, damping=0.3, callback=T.EarlyStoppingEP(max_increase=np.inf),
               initializer=T.CustomInit(a_init=[("x", "bwd", 0.1)], b_init=[("x", "bwd", b0)]))
    W = student.factors[1].W
    fs = [orc.GBPriorFactor((2, N), rho=0.5), orc.ComplexLinearFactor(W), orc.ModulusLikFactor(sample["y"])]
    with np.errstate(all="ignore"):
        res = orc.ep_chain(fs, [(2, N), (2, 192)], max_iter=300, damping=0.3, max_increase=np.inf,
                           init=lambda i, d: (0.1, b0) if (i, d) == (0, "bwd") else (0, 0.0))
    assert res["status"] == "converged" and abs(ep.n_iter - res["n_iter"]) <= 1
    r = ep.get_variables_data()["x"]["r"].reshape(2, N)
    close(r, res["r"][0], 1e-5)
    xh, x0 = r[0] + 1j * r[1], sample["x"][0] + 1j * sample["x"][1]
    ph = np.vdot(xh, x0)
    ph /= abs(ph)
    assert np.mean(np.abs(xh * ph - x0)**2) < 1e-6 * np.mean(np.abs(x0)**2)
    # the reference's way of scoring it (examples/glm/plot_complex_pr.py: TrackErrors with the phase-symmetric mse, algos/metrics.py:18-26):
    # a grid of 100 phases, so the score is bounded by the grid's resolution; the plain mse sees the arbitrary global phase
    track = T.TrackErrors({"x": sample["x"]}, metrics=["mse", "phase_mse", "sign_mse", "overlap"], every=10)
    ep2 = T.ExpectationPropagation(student, ctx=ctx)
    ep2.iterate(max_iter=300, damping=0.3, callback=T.JoinCallback([track, T.EarlyStoppingEP(max_increase=np.inf)]),
                initializer=T.CustomInit(a_init=[("x", "bwd", 0.1)], b_init=[("x", "bwd", b0)]))
    last = track.records[-1]
    assert set(last) == {"id", "iter", "mse", "phase_mse", "sign_mse", "overlap"} and last["id"] == "x"
    # (the last record is the last multiple of ten iterations, r the final iterate: both sit on the phase grid's floor)
    assert last["phase_mse"] == pytest.approx(T.phase_symmetric_mse(sample["x"], r), rel=1e-2)
    assert last["phase_mse"] < 2e-3 * np.mean(sample["x"]**2) and last["phase_mse"] < 0.05 * last["mse"]
    assert track.records[0]["phase_mse"] > 10 * last["phase_mse"]


def test_modulus_se_error_and_phase_retrieval_state_evolution(ctx, golden_complex):
    """ModulusLikelihood.compute_backward_error (2-D Bessel-weighted average) against the reference's dblquad values, and
    the StateEvolution recursion of phase retrieval (glm_state_evolution with a Marchenko-Pastur channel) against the
    reference's trajectory - every scalar message of every iteration"""
    import tramp_amd as T
    lk = T.ModulusLikelihood(y=None, ctx=ctx)
    for c in golden_complex["se"]:
        assert lk.compute_backward_error(c["az"], c["tau_z"]) == pytest.approx(c["v"], rel=1e-7), (c["az"], c["tau_z"])
        assert lk.compute_backward_state_evolution(c["az"], c["tau_z"]) == pytest.approx(c["a_new"], rel=1e-6)
    az = np.array([c["az"] for c in golden_complex["se"]])
    tz = np.array([c["tau_z"] for c in golden_complex["se"]])
    close(lk.compute_backward_error(az, tz), [c["v"] for c in golden_complex["se"]], 1e-7)
    for c in golden_complex["se_traj"]:
        model = T.glm_state_evolution(alpha=c["alpha"], prior_type="gauss_bernoulli", output_type="modulus",
                                      prior_rho=c["rho"])
        se = T.StateEvolution(model)
        rec = []
        stop = T.EarlyStopping()

        def cb(algo, i, max_iter):
            rec.append((algo.get_edges_data(), algo.get_variables_data()))
            return stop(algo, i, max_iter)
        se.iterate(max_iter=60, callback=cb, initializer=T.ConstantInit(a=c["a0"]))
        assert abs(se.n_iter - c["n_iter"]) <= 1
        t = c["traj"]
        for it in range(min(len(rec), t["v:x"].shape[0])):
            rt = 1e-5 if it < 8 else 1e-3           # quadrature differences compound over the recursion
            for vid in ("x", "z"):
                a_tot = t[f"fwd:{vid}:a"][it] + t[f"bwd:{vid}:a"][it]
                close(rec[it][0][(vid, "fwd")], t[f"fwd:{vid}:a"][it], rt, scale=a_tot)
                close(rec[it][0][(vid, "bwd")], t[f"bwd:{vid}:a"][it], rt, scale=a_tot)
                close(rec[it][1][vid]["v"], t[f"v:{vid}"][it], rt)


@pytest.mark.parametrize("which", ["phase_retrieval", "mixture", "exponential"])
def test_ensembles_and_graph_replay_of_the_round6_factors(ctx, which):
    """several instances at once (each must equal the oracle's single run on its own data), the device-side stopping rule,
    and the iteration captured as a hipGraph: the same bits as the eager loop"""
    import tramp_amd as T
    rs = np.random.RandomState(11)
    B = 3
    if which == "phase_retrieval":
        N, M = 40, 120
        W = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))) / np.sqrt(N)
        X = rs.standard_normal((B, 2, N)) * rs.binomial(1, 0.5, (B, 2, N))
        Y = np.abs(np.einsum("mn,bn->bm", W, X[:, 0] + 1j * X[:, 1]))
        b0 = rs.standard_normal((2, N))
        model = (T.GaussBernoulliPrior(size=(2, N), rho=0.5, ctx=ctx) @ T.SISOVariable(id="x")
                 @ T.ComplexLinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z") @ T.ModulusLikelihood(y=Y, ctx=ctx)).to_model()
        init = T.CustomInit(a_init=[("x", "bwd", 0.1)], b_init=[("x", "bwd", b0)])
        kw = dict(max_iter=300, damping=0.3, callback=T.EarlyStoppingEP(max_increase=np.inf), initializer=init)

        def oracle(b):
            fs = [orc.GBPriorFactor((2, N), rho=0.5), orc.ComplexLinearFactor(W), orc.ModulusLikFactor(Y[b])]
            return orc.ep_chain(fs, [(2, N), (2, M)], max_iter=300, damping=0.3, max_increase=np.inf,
                                init=lambda i, d: (0.1, b0) if (i, d) == (0, "bwd") else (0, 0.0))
        shape = (2, N)
    else:
        N, M, nv = 96, 64, 1e-2
        W = rs.randn(M, N) / np.sqrt(N)
        if which == "mixture":
            kwp = dict(probs=[0.2, 0.5, 0.3], means=[-2.0, 0.0, 1.5], vars=[0.3, 0.01, 2.0])
            prior, pf = T.GaussianMixturePrior(size=N, ctx=ctx, **kwp), orc.GMPriorFactor(N, kwp["probs"], kwp["means"], kwp["vars"])
            k = rs.choice(3, size=(B, N), p=kwp["probs"])
            X = np.asarray(kwp["means"])[k] + np.sqrt(np.asarray(kwp["vars"]))[k] * rs.standard_normal((B, N))
        else:
            prior, pf = T.ExponentialPrior(size=N, mean=0.7, ctx=ctx), orc.ExpPriorFactor(N, 0.7)
            X = rs.exponential(0.7, (B, N))
        Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((B, M))
        model = (prior @ T.SISOVariable(id="x") @ T.LinearChannel(W, ctx=ctx, svd="host") @ T.SISOVariable(id="z")
                 @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
        a0 = 1e-3 if which == "exponential" else 0
        kw = dict(max_iter=100, damping=0.2, initializer=T.CustomInit(a=a0, b=0))

        def oracle(b):
            fs = [pf, orc.LinearFactor(W), orc.GaussLikFactor(Y[b], nv)]
            return orc.ep_chain(fs, [N, M], max_iter=100, damping=0.2, init=lambda i, d: (a0, 0.0))
        shape = (N,)
    runs = []
    for graph in (False, True):
        ep = T.ExpectationPropagation(model, ctx=ctx)
        assert ep.batch == B
        ep.iterate(graph=graph, **kw)
        d = ep.get_variables_data()
        runs.append((ep.n_iter_instance.copy(), ep.status.copy(), np.asarray(d["x"]["r"]).copy(), np.asarray(d["x"]["v"]).copy()))
    for a, b in zip(*runs):
        assert np.array_equal(a, b), "the captured iteration differs from the eager loop"
    n_it, status, r, v = runs[0]
    code = dict(max_iter=0, converged=1, diverged=2)
    for b in range(B):
        with np.errstate(all="ignore"):
            res = oracle(b)
        # per instance: the oracle's own stopping decision (converged / gave up / cut short) and iteration count
        assert status[b] == code[res["status"]] and abs(int(n_it[b]) - res["n_iter"]) <= 1, (b, status[b], n_it[b], res["status"], res["n_iter"])
        close(r.reshape((B,) + shape)[b], res["r"][0], 2e-5)
        close(np.atleast_1d(v)[b], res["v"][0], 1e-4)
    if which == "phase_retrieval":
        assert np.all(status == 1)


@pytest.mark.parametrize("n", [3000, 1 << 18])
def test_mixture_narrow_component_far_from_zero_f32(ctx, n):
    "mean^2 / var = 4e4: the constants are formed without the cancellation between eta_k and normal.A (tramp_math.h: gm_const_k)"
    import tramp_amd as T
    rs = np.random.RandomState(n % 101)
    kw = dict(probs=[0.3, 0.7], means=[2.0, -0.5], vars=[1e-4, 0.5])
    B = 3
    ax = np.array([0.05, 1.3, 40.0])
    bx = (ax[:, None] * np.where(rs.rand(B, n) < 0.3, 2.0, -0.5) + np.sqrt(ax)[:, None] * rs.standard_normal((B, n))).astype(np.float32)
    r, v = T.GaussianMixturePrior(size=n, dtype=np.float32, ctx=ctx, **kw).compute_forward_posterior(ax, bx)
    for b in range(B):
        r_o, v_o = orc.gm_prior_posterior(ax[b], bx[b].astype(np.float64), kw["probs"], kw["means"], kw["vars"])
        close(r[b], r_o, 2e-5)
        close(v[b], v_o, 2e-4)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,kw", [("exponential", dict(mean=0.7)), ("positive", {}),
                                     ("mixture", dict(probs=[0.5, 0.5], means=[-1, 1], vars=[1, 1])),
                                     ("mixture", dict(probs=[0.2, 0.5, 0.3], means=[-2.0, 0.0, 1.5], vars=[0.3, 0.01, 2.0]))])
@pytest.mark.parametrize("damping", [0.0, 0.3])
def test_round6_priors_on_the_resident_pipeline(ctx, name, kw, dtype, damping):
    """many instances of several tiles each: the update takes the resident single-pass pipeline (the half-line priors its
    lean form, with the tile kernels' redo of the instances whose messages leave the erfcx table) - asserted - and equals
    the oracle instance by instance, with the damping fused; a non-finite message poisons only its own instance"""
    import tramp_amd as T
    B, n = 300, 7 * (2048 if dtype == np.float64 else 4096)
    if not ctx.resident_covers(dtype, n, B):
        pytest.skip("the pipeline does not cover this shape on this device")
    rs = np.random.RandomState(B + len(kw))
    ax = np.exp(rs.uniform(np.log(0.3), np.log(8.0), B))
    bx = (ax[:, None] * rs.standard_normal((B, n)) * 1.2 + 0.3).astype(dtype)
    bx[5, 100] = -60.0                 # far beyond the erfcx table for the half-line priors: this instance is redone
    bx[7, 9] = np.nan
    a_old, b_old = np.full(B, 0.4), (0.1 * rs.standard_normal((B, n))).astype(dtype)
    pr = make_prior(T, ctx, name, kw, n, dtype)
    post, _, _ = prior2_oracle(orc, name, kw)
    a_d, b_d = ctx.empty((B,), np.float64), ctx.empty((B, n), dtype)
    runs0 = ctx.resident_runs()
    extra = dict(damping=damping, old_a=ctx.array(a_old), old_b=ctx.array(b_old, dtype)) if damping else {}
    pr.compute_forward_message(ctx.array(ax), ctx.array(bx, dtype), out_a=a_d, out_b=b_d, **extra)
    assert ctx.resident_runs() == runs0 + 1, "the update did not take the resident pipeline"
    a_h, b_h = a_d.get(), b_d.get()
    rt = RT[dtype]
    assert np.isnan(a_h[7]) and np.isnan(b_h[7]).any() and np.isfinite(np.delete(a_h, 7)).all()
    for b in (0, 5, 6, 8, 150, 299):
        ro, vo = post(ax[b], bx[b].astype(np.float64))
        an, bn = orc.ab_new(ro, vo, ax[b], bx[b].astype(np.float64))
        if damping:
            an, bn = damping * a_old[b] + (1 - damping) * an, damping * b_old[b].astype(np.float64) + (1 - damping) * bn
        close(a_h[b], an, max(rt, 1e-7), scale=an + ax[b])
        close(b_h[b], bn, max(rt, 1e-7), scale=max(np.abs(bn).max(), np.abs(bx[b]).max()))


def test_phase_retrieval_ep_fp32(ctx, golden_complex):
    "fp32 vectors (scalars stay f64): 1e-3 on the posterior means of the reference's first iterations"
    import tramp_amd as T
    c = golden_complex["ep"][0]
    W = redraw_complex_W(c)
    N = c["N"]
    dt = np.float32
    model = (T.GaussBernoulliPrior(size=(2, N), rho=c["rho"], dtype=dt, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.ComplexLinearChannel(W, dtype=dt, ctx=ctx, svd="host") @ T.SISOVariable(id="z")
             @ T.ModulusLikelihood(y=c["y"], dtype=dt, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    rec = Recorder(T.PassCallback())
    ep.iterate(max_iter=5, callback=rec, damping=c["damping"],
               initializer=T.CustomInit(a_init=[("x", "bwd", c["a0"])], b_init=[("x", "bwd", c["b0"])], a=0, b=0))
    t = c["traj"]
    for it in range(5):
        for vid in ("x", "z"):
            ref = t[f"r:{vid}"][it]
            close(np.asarray(rec.iters[it]["vars"][vid]["r"]).reshape(ref.shape), ref, 2e-3, scale=max(np.abs(ref).max(), 1e-3))
            close(rec.iters[it]["vars"][vid]["v"], t[f"v:{vid}"][it], 2e-3)


def test_teacher_student_scenarios_and_critical_alpha(ctx):
    """tramp/experiments: BayesOptimalScenario.run_all on the sparse GLM (EP's variance and mse agree with state evolution), the same
    with an ensemble of teacher samples as ONE run, the convergence data frames, and find_critical_alpha on noiseless compressed
    sensing (perfect recovery above the Gauss-Bernoulli threshold, alpha_c between rho and 1)"""
    import tramp_amd as T
    np.random.seed(7)
    N, alpha, rho = 2000, 0.8, 0.2
    model = T.glm_generative(N=N, alpha=alpha, ensemble_type="gaussian", prior_type="gauss_bernoulli", output_type="gaussian",
                             prior_rho=rho, output_var=1e-3, ctx=ctx)
    sc = T.BayesOptimalScenario(model, x_ids=["x"])
    recs = sc.run_all(metrics=["mse", "overlap"], max_iter=100, damping=0.1)
    by = {r["source"]: r for r in recs}
    assert set(by) == {"SE", "EP", "mse", "overlap"} and by["EP"]["n_iter"] < 100
    assert by["EP"]["v"] == pytest.approx(by["mse"]["v"], rel=0.3) and by["EP"]["v"] == pytest.approx(by["SE"]["v"], rel=0.3)
    assert by["mse"]["v"] < 0.05 * rho
    # four teacher samples on the ensemble axis: one EP run, per-instance scores
    recs4 = sc.run_all(source="EP", metrics=["mse"], instances=4, max_iter=100, damping=0.1)
    by4 = {r["source"]: r for r in recs4}
    assert sc.ep.batch == 4 and np.shape(by4["mse"]["v"]) == (4,) and np.shape(by4["EP"]["v"]) == (4,)
    assert np.all(np.asarray(by4["mse"]["v"]) < 0.05 * rho) and np.ptp(by4["mse"]["v"]) > 0
    np.testing.assert_allclose(by4["mse"]["v"], by4["EP"]["v"], rtol=0.3)
    # convergence data frames
    sc.setup(seed=3)
    df = sc.ep_convergence(metrics=["mse"], max_iter=30, damping=0.1)
    assert list(df.columns) == ["id", "iter", "mse", "v"] and len(df) >= 5 and float(df["mse"].iloc[-1]) < float(df["mse"].iloc[0])
    dfs = sc.se_convergence(max_iter=30)
    # (the damped run of run_all stops on its tolerance a little short of the fixed point)
    assert float(dfs["v"].iloc[-1]) == pytest.approx(by["SE"]["v"], rel=1e-2)
    assert T.run_state_evolution(["x"], sc.student, max_iter=30)[0]["v"] == pytest.approx(float(dfs["v"].iloc[-1]), rel=1e-2)
    # critical alpha of noiseless sparse recovery from an informed start: between rho (counting bound) and 1
    ac = T.find_critical_alpha("x", 1e3, "perfect", 0.05, 0.95, T.glm_state_evolution, alpha_tol=1e-2, vtol=1e-6,
                               prior_type="gauss_bernoulli", output_type="gaussian", prior_rho=0.3, output_var=1e-11)
    assert 0.3 <= ac < 0.6
    v_above = T.find_state_evolution_mse("x", 1e3, ac + 0.05, T.glm_state_evolution, prior_type="gauss_bernoulli",
                                         output_type="gaussian", prior_rho=0.3, output_var=1e-11)
    v_below = T.find_state_evolution_mse("x", 1e3, max(ac - 0.05, 0.05), T.glm_state_evolution, prior_type="gauss_bernoulli",
                                         output_type="gaussian", prior_rho=0.3, output_var=1e-11)
    assert v_above < 1e-6 < v_below
