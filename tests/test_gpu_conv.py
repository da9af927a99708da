"""GPU parity of the convolution channel family (csrc/conv.hip) against
vectors captured from the reference's ConvChannel / Blur / Differential / Laplacian channels,
the oracle, and size-independent properties at n = 10^6."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _own_fft_engine(monkeypatch):
    """These tests exercise the own FFT engine at sizes below the default switch-over (csrc/conv.hip own_engine_wanted:
    rocFFT under 32768 points per signal until round 5; now the default anyway); tests that compare engines set TRAMP_CONV_FFT themselves."""
    monkeypatch.setenv("TRAMP_CONV_FFT", "own")

TOL = {np.float64: 1e-10, np.float32: 2e-4}


def close(a, b, rtol, scale=None, atol=0.0):
    b = np.asarray(b)
    sc = np.abs(b).max() if scale is None else scale
    np.testing.assert_allclose(np.asarray(a), b, rtol=rtol, atol=rtol * max(sc, 1e-300) + atol)


class Recorder:
    def __init__(self, stopper):
        self.stopper = stopper
        self.iters = []

    def __call__(self, algo, i, max_iter):
        rec = dict(edges=algo.get_edges_data(), vars=algo.get_variables_data())
        rec["b"] = {(v.id, d): algo.get_message(v.id, d)[1] for v in algo.variables for d in ("fwd", "bwd")}
        self.iters.append(rec)
        return self.stopper(algo, i, max_iter)


def make_channel(T, ctx, c, dtype=np.float64):
    kw = dict(c["kwargs"])
    return T.get_channel(c["kind"], dtype=dtype, ctx=ctx, **kw)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_conv_channel_matches_reference(ctx, golden_conv, dtype):
    import tramp_amd as T
    rt = TOL[dtype]
    for c in golden_conv["channels"]:
        ch = make_channel(T, ctx, c, dtype)
        assert tuple(ch.shape) == c["filter"].shape
        close(ch.filter, c["filter"], 1e-12)
        close(ch.second_moment(1.7), c["second_moment"], 1e-12)
        close(ch.sample(c["z"]), c["x"], rt)
        o = orc.ConvOracle(c["filter"])
        hs = np.fft.rfftn(c["filter"])
        close(ch.half_spectrum, (np.abs(hs)**2).reshape(-1), 1e-12)
        for k in c["cases"]:
            az, ax = k["az"], k["ax"]
            if az < 1e-100:
                continue           # az/ax ~ 1e-301: n_eff decided by whether a spectrum zero is exact
            if dtype == np.float32 and (az < 1e-6 or ax > 1e3 or az > 1e6):
                continue           # fp32 vectors cannot carry 1e11-conditioned spectral solves
            rz, vz = ch.compute_backward_posterior(az, c["bz"], ax, c["bx"])
            rx, vx = ch.compute_forward_posterior(az, c["bz"], ax, c["bx"])
            assert rz.shape == c["bz"].shape and rx.shape == c["bx"].shape
            close(rz, k["rz"], rt)
            # rx = W rz: a spectrum entry that is an exact 0 in NumPy's FFT (DC of a difference
            # stencil) is ~1e-17 in another FFT; with az -> 0 that bin of rz is ~1/az
            rx_atol = 64 * 2.3e-16 * np.abs(k["rz"]).max() * np.sqrt(c["spectrum"].max())
            close(rx, k["rx"], rt, atol=rx_atol)
            # vz = (1 - n_eff) / az cancels when n_eff -> 1 (az << ax s): one ulp of n_eff is
            # eps / az absolute on vz; tiny spectra (blur filters) make n_eff itself 1e-12-conditioned
            vz_cond = 32 * 2.3e-16 / az / max(k["vz"], 1e-300) + (1e-9 if c["kind"].startswith("blur") else 0)
            close(vz, k["vz"], 1e-12 + vz_cond)
            close(vx, k["vx"], 1e-11)
            close(ch.compute_n_eff(az, ax), k["n_eff"], 1e-11, scale=1.0)
            # a_new = 1/v - a and b_new = r (a + a_new) - b cancel: compare on the scale of the terms
            af, bf = ch.compute_forward_message(az, c["bz"], ax, c["bx"])
            ab, bb = ch.compute_backward_message(az, c["bz"], ax, c["bx"])
            close(af, k["a_fwd"], 1e-10, scale=abs(ax) + 1 / max(k["vx"], 1e-300))
            close(ab, k["a_bwd"], 1e-10 + vz_cond, scale=abs(az) + 1 / max(k["vz"], 1e-300))
            close(bf, k["b_fwd"], rt, scale=np.abs(k["rx"]).max() * (ax + k["a_fwd"]) + np.abs(c["bx"]).max(),
                  atol=rx_atol * (ax + k["a_fwd"]))
            close(bb, k["b_bwd"], rt + vz_cond,
                  scale=np.abs(k["rz"]).max() * (az + k["a_bwd"]) + np.abs(c["bz"]).max())
            # the oracle agrees too (it is what larger cases are checked against)
            close(rz, o.backward_mean(az, c["bz"], ax, c["bx"]), rt)


def test_conv_variance_edge_cases_and_batch(ctx):
    import tramp_amd as T
    rs = np.random.RandomState(3)
    f = rs.randn(5, 12) / 5
    ch = T.ConvChannel(f, ctx=ctx)
    o = orc.ConvOracle(f)
    az = np.array([1.3, 2.0, 0.0, 1e11, 7.0, 1e-3])
    ax = np.array([0.7, 0.0, 5.0, 1e-11, 7e5, 2.0])
    v_f, a_f, ne = ch.device_variance(T._lib.FWD, az, ax)
    v_b, a_b, _ = ch.device_variance(T._lib.BWD, az, ax)
    for i in range(len(az)):
        close(ne[i], o.n_eff(az[i], ax[i]), 1e-13, scale=1.0)
        close(v_f[i], o.forward_variance(az[i], ax[i]), 1e-13)
        close(a_f[i], orc.a_new(o.forward_variance(az[i], ax[i]), ax[i]), 1e-11, scale=ax[i] + 1 / v_f[i])
        if az[i] > 0:
            close(v_b[i], o.backward_variance(az[i], ax[i]), 1e-13)
            close(a_b[i], orc.a_new(o.backward_variance(az[i], ax[i]), az[i]), 1e-11, scale=az[i] + 1 / v_b[i])
    assert ne[1] == 0.0 and ne[2] == 1.0                  # conv_channel.py:71-76
    # ensemble rows are independent instances with their own (az, ax)
    B = 4
    bz = rs.standard_normal((B,) + f.shape)
    bx = rs.standard_normal((B,) + f.shape)
    azb, axb = np.array([1.3, 0.4, 9.0, 2.2]), np.array([0.7, 3.0, 0.1, 2.2])
    a_new, b_new = ch.compute_backward_message(azb, bz, axb, bx)
    assert b_new.shape == bz.shape
    for b in range(B):
        a1, b1 = o.backward_message(azb[b], bz[b], axb[b], bx[b])
        close(a_new[b], a1, 1e-11, scale=azb[b] + a1)
        close(b_new[b], b1, 1e-10)


def build_deconv(T, ctx, c, dtype=np.float64, y=None):
    shape = tuple(c["shape"])
    model = (T.GaussBernoulliPrior(size=shape, rho=c["rho"], dtype=dtype, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.ConvChannel(c["filter"], dtype=dtype, ctx=ctx) @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=(c["y"] if y is None else y).reshape(-1, int(np.prod(shape))),
                                    var=c["noise_var"], dtype=dtype, ctx=ctx)).to_model()
    return model


def test_conv_ep_trajectories_match_reference(ctx, golden_conv):
    import tramp_amd as T
    for c in golden_conv["ep"]:
        model = build_deconv(T, ctx, c)
        ep = T.ExpectationPropagation(model, ctx=ctx)
        rec = Recorder(T.EarlyStoppingEP())
        ep.iterate(max_iter=40, callback=rec, damping=c["damping"])
        assert ep.n_iter == c["n_iter"], c["name"]
        t = c["traj"]
        n = int(np.prod(c["shape"]))
        for it in range(min(t["v:x"].shape[0], len(rec.iters))):
            rt = 1e-7 if it < 6 else 1e-5
            for vid in ("x", "z"):
                e = rec.iters[it]["edges"]
                a_tot = t[f"fwd:{vid}:a"][it] + t[f"bwd:{vid}:a"][it]
                close(e[(vid, "fwd")][0], t[f"fwd:{vid}:a"][it], rt, scale=a_tot)
                close(e[(vid, "bwd")][0], t[f"bwd:{vid}:a"][it], rt, scale=a_tot)
                close(rec.iters[it]["b"][(vid, "fwd")].reshape(n), t[f"fwd:{vid}:b"][it].reshape(n), rt)
                close(rec.iters[it]["b"][(vid, "bwd")].reshape(n), t[f"bwd:{vid}:b"][it].reshape(n), rt)
                close(np.asarray(rec.iters[it]["vars"][vid]["r"]).reshape(n), t[f"r:{vid}"][it].reshape(n), rt)
                close(rec.iters[it]["vars"][vid]["v"], t[f"v:{vid}"][it], rt)
        close(np.asarray(ep.get_variables_data()["x"]["r"]).reshape(n), c["final_r"].reshape(n), 1e-5)


@pytest.mark.parametrize("damping", [None, 0.2])
def test_conv_ep_graph_replay_and_ensemble(ctx, golden_conv, damping):
    """hipGraph replay of a conv iteration == eager; every instance of an ensemble == oracle EP."""
    import tramp_amd as T
    c = golden_conv["ep"][0]
    n = int(np.prod(c["shape"]))
    rs = np.random.RandomState(8)
    o = orc.ConvOracle(c["filter"])
    B = 3
    X = np.stack([orc.gb_sample(rs, n, c["rho"]) for _ in range(B)])
    Y = o.convolve(X) + np.sqrt(c["noise_var"]) * rs.standard_normal((B, n))
    res = {}
    for graph in (False, True):
        ep = T.ExpectationPropagation(build_deconv(T, ctx, c, y=Y), ctx=ctx)
        ep.iterate(max_iter=9, callback=T.PassCallback(), damping=damping, graph=graph)
        res[graph] = np.asarray(ep.get_variables_data()["x"]["r"]).reshape(B, n)
    assert np.array_equal(res[False], res[True])
    for b in range(B):
        fs = [orc.GBPriorFactor(n, rho=c["rho"]), orc.ConvFactor(c["filter"]),
              orc.GaussLikFactor(Y[b], c["noise_var"])]
        ref = orc.ep_chain(fs, [n, n], max_iter=9, damping=damping, tol=0.0)
        close(res[False][b], ref["r"][0], 1e-6)


@pytest.mark.parametrize("dtype,n", [(np.float64, 1_000_000), (np.float32, 1 << 20)])
def test_conv_large_n_properties(ctx, dtype, n):
    """n = 10^6 (SURVEY config 2's N): size-independent identities of the update.
    rx = W rz; az rz + ax W^T rx' ... reduces to the normal equations (az + ax W^T W) rz = bz + W^T bx,
    checked in the Fourier domain on the host with one rfft of the device result."""
    import tramp_amd as T
    rs = np.random.RandomState(1)
    f = np.zeros(n)
    f[:64] = rs.randn(64) / 8                              # short random blur kernel
    ch = T.ConvChannel(f, dtype=dtype, ctx=ctx)
    close(ch.spectrum_mean, np.sum(f**2), 1e-11)           # Parseval
    bz = rs.standard_normal(n)
    bx = rs.standard_normal(n)
    az, ax = 1.7, 3.1
    rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
    rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
    rt = 1e-10 if dtype == np.float64 else 1e-4
    close(rx, ch.convolve(rz), rt)                         # conv_channel.py:105-109
    wbar = np.fft.rfft(f)
    lhs = (az + ax * np.abs(wbar)**2) * np.fft.rfft(rz)
    rhs = np.fft.rfft(bz) + wbar * np.fft.rfft(bx)
    close(lhs, rhs, rt)
    close(az * vz + ax * vx, 1.0, 1e-12)                   # (1 - n_eff) + n_eff
    a_new, b_new = ch.compute_backward_message(az, bz, ax, bx)
    close(b_new, rz * (az + a_new) - bz, rt * 10)


@pytest.mark.parametrize("dtype,n", [(np.float64, 1 << 13), (np.float64, 1 << 16), (np.float32, 1 << 14),
                                     (np.float64, 20000), (np.float64, 100000), (np.float32, 25600)])
def test_conv_own_fft_engine_matches_oracle(ctx, dtype, n):
    """1-D sizes n = 2 h, h = 2^a 5^b, run on the hand-written two-pass FFT (csrc/fft_pow2.hip) with the spectrum
    kept in the transform's permuted order: posteriors, messages, samples and a damped EP run (which also
    exercises the affine update of the cached transform) against the oracle."""
    import tramp_amd as T
    rs = np.random.RandomState(n % 1000)
    f = np.zeros(n)
    f[:48] = rs.randn(48) / 6
    f[-5:] = rs.randn(5) / 6
    ch = T.ConvChannel(f, dtype=dtype, ctx=ctx)
    o = orc.ConvOracle(f)
    rt = 1e-10 if dtype == np.float64 else 3e-4
    B = 2
    bz, bx = rs.standard_normal((B, n)), rs.standard_normal((B, n)) * 2
    az, ax = np.array([1.7, 0.3]), np.array([3.1, 5.0])
    rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
    rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
    for b in range(B):
        close(rz[b], o.backward_mean(az[b], bz[b], ax[b], bx[b]), rt)
        close(rx[b], o.forward_mean(az[b], bz[b], ax[b], bx[b]), rt)
        close(vz[b], o.backward_variance(az[b], ax[b]), 1e-11)
        close(vx[b], o.forward_variance(az[b], ax[b]), 1e-11)
    close(ch.convolve(bz[0]), o.convolve(bz[0]), rt)
    a_new, b_new = ch.compute_backward_message(az, bz, ax, bx)
    for b in range(B):
        a1, b1 = o.backward_message(az[b], bz[b], ax[b], bx[b])
        close(a_new[b], a1, 1e-10, scale=az[b] + a1)
        close(b_new[b], b1, rt * 10)
    if dtype == np.float64:
        rho, nv, d = 0.05, 1e-3, 0.2
        x = orc.gb_sample(rs, n, rho)
        y = o.convolve(x) + np.sqrt(nv) * rs.standard_normal(n)
        model = (T.GaussBernoulliPrior(size=n, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ ch @ T.SISOVariable(id="z")
                 @ T.GaussianLikelihood(y=y, var=nv, ctx=ctx)).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx)
        ep.iterate(max_iter=6, callback=T.PassCallback(), damping=d)
        fs = [orc.GBPriorFactor(n, rho=rho), orc.ConvFactor(f), orc.GaussLikFactor(y, nv)]
        ref = orc.ep_chain(fs, [n, n], max_iter=6, damping=d, tol=0.0)
        r_eager = np.asarray(ep.get_variables_data()["x"]["r"]).reshape(n)
        close(r_eager, ref["r"][0], 1e-6)
        # the own transforms capture into a hipGraph like any other launch: replay == eager, bit for bit
        ep2 = T.ExpectationPropagation(model, ctx=ctx)
        ep2.iterate(max_iter=6, callback=T.PassCallback(), damping=d, graph=True)
        assert np.array_equal(np.asarray(ep2.get_variables_data()["x"]["r"]).reshape(n), r_eager)


def test_conv_state_evolution_matches_oracle(ctx):
    """SE of the deconvolution model: the channel's errors are its posterior variances (conv_channel.py:137-143),
    evaluated by the device n_eff reduction; the prior's by the device quadrature."""
    import tramp_amd as T
    rs = np.random.RandomState(4)
    f = rs.randn(4096) / 64
    rho, nv = 0.1, 1e-2
    model = (T.GaussBernoulliPrior(size=None, rho=rho, ctx=ctx) @ T.SISOVariable(id="x")
             @ T.ConvChannel(f, ctx=ctx) @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=None, var=nv, ctx=ctx)).to_model()
    se = T.StateEvolution(model)
    recs = []
    se.iterate(max_iter=12, callback=lambda a, i, m: recs.append(a.get_variables_data()) and False)
    fs = [orc.GBPriorFactor(None, rho=rho), orc.ConvFactor(f), orc.GaussLikFactor(None, nv)]
    ref = []
    orc.se_chain(fs, max_iter=12, tol=0.0, method="gl", record=lambda it, fwd, bwd, vs: ref.append(list(vs)))
    assert len(recs) == len(ref) == 12
    for it in range(12):
        close(recs[it]["x"]["v"], ref[it][0], 1e-5)
        close(recs[it]["z"]["v"], ref[it][1], 1e-5)
