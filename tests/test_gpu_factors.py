"""GPU parity: HIP factor kernels (through the C ABI) vs the CPU oracle and the
golden vectors captured from the reference.

fp64: 1e-9 relative (the north-star bar is 1e-6), fp32: 1e-3; masks/counts bit-exact.
"""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu

F64 = dict(rtol=1e-9, atol_scale=1e-10)
F32 = dict(rtol=1e-3, atol_scale=1e-4)     # the north star's fp32 budget: 1e-3 on posterior means


def tol(dtype):
    return F64 if dtype == np.float64 else F32


def close(a, b, dtype=np.float64, scale=None, atol=0.0):
    t = tol(dtype)
    b = np.asarray(b, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    sc = np.nanmax(np.abs(b)) if scale is None and b.size else (scale or 0.0)
    np.testing.assert_allclose(a, b, rtol=t["rtol"], atol=t["atol_scale"] * max(sc, 1e-300) + atol,
                               equal_nan=True)


EPS = {np.float64: 2.3e-16, np.float32: 1.2e-7}


def close_a(a_new, ref, a_in, dtype=np.float64):
    """a_new = clip(1/v - a_in): conditioned on 1/v = a_in + a_new (scalars are always f64,
    but v carries the vector dtype's reduction error)."""
    cond = (abs(a_in) + abs(ref)) * (64 * EPS[np.float64] + tol(dtype)["rtol"] * (dtype == np.float32))
    close(a_new, ref, dtype, atol=cond)


def close_b(b_new, ref, r_ref, a_in, a_new_ref, b_in, dtype=np.float64):
    "b_new = r (a_in + a_new) - b_in: conditioned on |r| (a_in + a_new) + |b_in|"
    sc = np.nanmax(np.abs(r_ref)) * (abs(a_in) + abs(a_new_ref)) + np.nanmax(np.abs(b_in))
    close(b_new, ref, dtype, scale=sc, atol=1e3 * EPS[dtype] * sc)


@pytest.fixture(scope="module")
def T():
    import tramp_amd
    return tramp_amd


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gb_prior_golden(T, ctx, golden_factors, dtype):
    ran = 0
    for c in golden_factors["gb"]:
        if dtype == np.float32 and (np.abs(c["b"]).max() > 1e4 or c["ax"] > 1e6 or c["ax"] < 1e-6):
            continue     # outside f32 range by construction
        ran += 1
        p = T.GaussBernoulliPrior(size=c["b"].shape[0], rho=c["rho"], mean=c["mean"], var=c["var"],
                                  dtype=dtype, ctx=ctx)
        r, v = p.compute_forward_posterior(c["ax"], c["b"])
        close(r, c["r"], dtype)
        close(v, c["v"], dtype)
        a_new, b_new = p.compute_forward_message(c["ax"], c["b"])
        close_a(a_new, c["a_new"], c["ax"], dtype)
        close_b(b_new, c["b_new"], c["r"], c["ax"], c["a_new"], c["b"], dtype)
        assert isinstance(a_new, float) and b_new.shape == c["b"].shape
    # a fixture change must not silently empty the loop: every case in f64, most of them in f32
    assert ran == len(golden_factors["gb"]) if dtype == np.float64 else ran >= max(3, len(golden_factors["gb"]) // 2), ran
    d = golden_factors["gb_diag"]
    p = T.GaussBernoulliPrior(size=100, rho=0.5, isotropic=False, dtype=dtype, ctx=ctx)
    r, v = p.compute_forward_posterior(d["ax"], d["bx"])
    close(r, d["r"], dtype); close(v, d["v"], dtype)
    a_new, b_new = p.compute_forward_message(d["ax"], d["bx"])
    close(a_new, d["a_new"], dtype); close(b_new, d["b_new"], dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_l1_prior_and_mask(T, ctx, golden_factors, dtype):
    ran = 0
    for c in golden_factors["l1"]:
        b = c["b"]
        if dtype == np.float32:
            b32 = b.astype(np.float32)
            if not np.all(np.isfinite(b32)):
                continue
            ref_mask = orc.l1_support_mask(b32, np.float32(c["gamma"]))
        else:
            ref_mask = c["mask"]
        p = T.MAP_L1NormPrior(size=b.shape[0], gamma=c["gamma"], dtype=dtype, ctx=ctx)
        mask, count = p.support(c["ax"], b)
        assert np.array_equal(mask, ref_mask)              # bit-exact row
        assert int(count[0]) == int(ref_mask.sum())
        if dtype == np.float64:
            r_ref, v_ref, a_ref, b_ref, b_in = c["r"], c["v"], c["a_new"], c["b_new"], b
        else:
            # f32: the reference's formulas on the f32-rounded message (the support then matches bit for bit)
            if np.abs(b32).max() > 1e4 or c["ax"] > 1e6 or c["ax"] < 1e-6:
                continue
            b_in = b32.astype(np.float64)
            r_ref, v_ref = orc.l1_prior_posterior(c["ax"], b_in, gamma=float(np.float32(c["gamma"])))
            a_ref, b_ref = orc.l1_prior_message(c["ax"], b_in, gamma=float(np.float32(c["gamma"])))
        r, v = p.compute_forward_posterior(c["ax"], b)
        close(r, r_ref, dtype); close(v, v_ref, dtype)
        a_new, b_new = p.compute_forward_message(c["ax"], b)
        close_a(a_new, a_ref, c["ax"], dtype)
        close_b(b_new, b_ref, r_ref, c["ax"], a_ref, b_in, dtype)
        ran += 1
    assert ran == len(golden_factors["l1"]) if dtype == np.float64 else ran >= max(2, len(golden_factors["l1"]) // 2), ran


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gaussian_prior_likelihood(T, ctx, golden_factors, dtype):
    c = golden_factors["gauss_prior"]
    p = T.GaussianPrior(size=c["b"].shape[0], mean=c["mean"], var=c["var"], dtype=dtype, ctx=ctx)
    a_new, b_new = p.compute_forward_message(c["ax"], c["b"])
    close(a_new, c["a_new"], dtype); close(b_new, c["b_new"], dtype)
    close(p.compute_forward_error(c["ax"]), c["err"])
    c = golden_factors["gauss_lik"]
    lk = T.GaussianLikelihood(y=c["y"], var=c["var"], dtype=dtype, ctx=ctx)
    a_new, b_new = lk.compute_backward_message(c["az"], c["b"])
    close(a_new, c["a_new"], dtype); close(b_new, c["b_new"], dtype)
    close(lk.compute_backward_error(c["az"], 1.0), c["err"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sgn_abs_likelihood(T, ctx, golden_factors, dtype):
    ran = 0
    for c in golden_factors["sgn_lik"]:
        if dtype == np.float32 and c["az"] < 1e-6:
            continue
        ran += 1
        lk = T.SgnLikelihood(y=c["y"], dtype=dtype, ctx=ctx)
        r, v = lk.compute_backward_posterior(c["az"], c["bz"], c["y"])
        sc = np.abs(c["r"]).max()
        close(r, c["r"], dtype, scale=sc); close(v, c["v"], dtype)
        a_new, b_new = lk.compute_backward_message(c["az"], c["bz"])
        close_a(a_new, c["a_new"], c["az"], dtype)
        close_b(b_new, c["b_new"], c["r"], c["az"], c["a_new"], c["bz"], dtype)
        if dtype == np.float64:
            _, vd = T.SgnLikelihood(y=c["y"], isotropic=False, ctx=ctx).compute_backward_posterior(
                c["az"], c["bz"], c["y"])
            close(vd, c["v_diag"], dtype)
    assert ran == len(golden_factors["sgn_lik"]) if dtype == np.float64 else ran >= max(2, len(golden_factors["sgn_lik"]) // 2), ran
    for c in golden_factors["abs_lik"]:
        lk = T.AbsLikelihood(y=c["y"], dtype=dtype, ctx=ctx)
        r, v = lk.compute_backward_posterior(c["az"], c["bz"], c["y"])
        close(r, c["r"], dtype); close(v, c["v"], dtype)
        a_new, b_new = lk.compute_backward_message(c["az"], c["bz"])
        close_a(a_new, c["a_new"], c["az"], dtype)
        close_b(b_new, c["b_new"], c["r"], c["az"], c["a_new"], c["bz"], dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_relu_likelihoods(T, ctx, golden_factors, dtype):
    for c in golden_factors["relu_lik"]:
        lk = (T.ReluLikelihood(y=c["y"], dtype=dtype, ctx=ctx) if c["slope"] == 0
              else T.LeakyReluLikelihood(slope=c["slope"], y=c["y"], dtype=dtype, ctx=ctx))
        r, v = lk.compute_backward_posterior(c["az"], c["bz"], c["y"])
        close(r, c["r"], dtype, scale=max(1.0, np.abs(c["r"]).max())); close(v, c["v"], dtype, atol=1e-300)
        a_new, b_new = lk.compute_backward_message(c["az"], c["bz"])
        close_a(a_new, c["a_new"], c["az"], dtype)
        close_b(b_new, c["b_new"], c["r"], c["az"], c["a_new"], c["bz"], dtype)
    # an observation no region contains poisons the message, as the reference (softmax of -inf, -inf)
    a_new, _ = T.ReluLikelihood(y=np.array([1.0, -0.5, 0.0]), ctx=ctx).compute_backward_message(1.0, np.zeros(3))
    assert np.isnan(a_new)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pwl_channels(T, ctx, golden_factors, dtype):
    cls = dict(sgn=T.SgnChannel, abs=T.AbsChannel, relu=T.ReluChannel)
    ran = 0
    for c in golden_factors["pwl"]:
        if dtype == np.float32 and (c["az"] < 1e-6 or c["ax"] != 0 and c["ax"] < 1e-6):
            continue
        if np.isnan(c["v_fwd"]) or np.isnan(c["v_bwd"]):
            continue           # reference itself overflows on this row
        ran += 1
        ch = cls[c["kind"]](dtype=dtype, ctx=ctx)
        for d, post, msg in (("fwd", ch.compute_forward_posterior, ch.compute_forward_message),
                             ("bwd", ch.compute_backward_posterior, ch.compute_backward_message)):
            r, v = post(c["az"], c["bz"], c["ax"], c["bx"])
            close(r, c["r_" + d], dtype, scale=max(1.0, np.abs(c["r_" + d]).max()))
            close(v, c["v_" + d], dtype)
            a_new, b_new = msg(c["az"], c["bz"], c["ax"], c["bx"])
            tgt = c["bx"] if d == "fwd" else c["bz"]
            a_in = c["ax"] if d == "fwd" else c["az"]
            close_a(a_new, c["a_" + d], a_in, dtype)
            # r = p0 r_pos + p1 r_neg cancels when the regions are symmetric: condition on the
            # region means, not on |r|
            with np.errstate(all="ignore"):
                rk = [orc.pwl_region_terms(sd, sl, x0, c["az"], c["bz"], c["ax"], c["bx"])[0] * (sl if d == "fwd" else 1)
                      + (x0 if d == "fwd" else 0) for sd, sl, x0 in orc.PWL_REGIONS[c["kind"]]]
            r_cond = np.maximum(np.abs(c["r_" + d]), np.minimum(np.abs(rk[0]), np.abs(rk[1])))
            close_b(b_new, c["b_" + d], r_cond, a_in, c["a_" + d], tgt, dtype)
    assert ran >= (18 if dtype == np.float64 else 12), ran         # of the 27 golden rows (some overflow in the reference itself)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pwl2_channels(T, ctx, golden_factors, dtype):
    "LeakyRelu / AsymmetricAbs through the generic K7 entry point (tramp_pwl2_channel)"
    for c in golden_factors["pwl2"]:
        if np.isnan(c["v_fwd"]) or np.isnan(c["v_bwd"]):
            continue
        ch = (T.LeakyReluChannel(slope=c["param"], dtype=dtype, ctx=ctx) if c["kind"] == "l-relu"
              else T.AsymmetricAbsChannel(shift=c["param"], dtype=dtype, ctx=ctx))
        close(ch.second_moment(1.7), c["tau_x"], atol=1e-12)
        for d, post, msg in (("fwd", ch.compute_forward_posterior, ch.compute_forward_message),
                             ("bwd", ch.compute_backward_posterior, ch.compute_backward_message)):
            r, v = post(c["az"], c["bz"], c["ax"], c["bx"])
            with np.errstate(all="ignore"):
                rk = [orc.pwl_region_terms(sd, sl, x0, c["az"], c["bz"], c["ax"], c["bx"], zb)[0] * (sl if d == "fwd" else 1)
                      + (x0 if d == "fwd" else 0) for sd, sl, x0, zb in orc.pwl_regions(c["kind"], c["param"])]
            r_cond = np.maximum(np.abs(c["r_" + d]), np.minimum(np.abs(rk[0]), np.abs(rk[1])))
            close(r, c["r_" + d], dtype, scale=max(1.0, np.nanmax(r_cond)))
            close(v, c["v_" + d], dtype)
            a_new, b_new = msg(c["az"], c["bz"], c["ax"], c["bx"])
            tgt = c["bx"] if d == "fwd" else c["bz"]
            a_in = c["ax"] if d == "fwd" else c["az"]
            close_a(a_new, c["a_" + d], a_in, dtype)
            close_b(b_new, c["b_" + d], r_cond, a_in, c["a_" + d], tgt, dtype)


def _pwl3(T, c, dtype, ctx):
    if c["kind"] == "h-tanh":
        return T.HardTanhChannel(dtype=dtype, ctx=ctx)
    if c["kind"] == "h-sigm":
        return T.HardSigmoidChannel(dtype=dtype, ctx=ctx)
    return T.SymmetricDoorChannel(width=c["param"], dtype=dtype, ctx=ctx)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pwl3_channels(T, ctx, golden_factors, dtype):
    "HardTanh / HardSigmoid / SymmetricDoor through tramp_pwl3_channel"
    for c in golden_factors["pwl3"]:
        if dtype == np.float32 and c["az"] >= 30:
            continue          # the pos/neg erfcx differences of the mid interval cancel beyond f32 there
        ch = _pwl3(T, c, dtype, ctx)
        close(ch.second_moment(1.7), c["tau_x"], atol=1e-12)
        for d, post, msg in (("fwd", ch.compute_forward_posterior, ch.compute_forward_message),
                             ("bwd", ch.compute_backward_posterior, ch.compute_backward_message)):
            r, v = post(c["az"], c["bz"], c["ax"], c["bx"])
            with np.errstate(all="ignore"):
                rk = []
                for zmin, zmax, slope, x0 in orc.pwl3_regions(c["kind"], c["param"]):
                    a = c["az"] + slope**2 * c["ax"]
                    b = c["bz"] + slope * (c["bx"] - c["ax"] * x0)
                    rz = orc._region_moments(zmin, zmax, b / a, 1 / a)[0]
                    rk.append(np.abs(slope * rz + x0 if d == "fwd" else rz))
            r_cond = np.maximum(np.abs(c["r_" + d]), np.sort(np.stack(rk), axis=0)[1])
            close(r, c["r_" + d], dtype, scale=max(1.0, np.nanmax(r_cond)))
            close(v, c["v_" + d], dtype, atol=(1e-9 if dtype == np.float64 else 3e-3) * c["v_" + d])
            a_new, b_new = msg(c["az"], c["bz"], c["ax"], c["bx"])
            tgt = c["bx"] if d == "fwd" else c["bz"]
            a_in = c["ax"] if d == "fwd" else c["az"]
            close_a(a_new, c["a_" + d], a_in, dtype)
            close_b(b_new, c["b_" + d], r_cond, a_in, c["a_" + d], tgt, dtype)


def test_pwl3_interval_branch_masks_bit_exact(T, ctx, golden_factors):
    "close / neg / pos / other of the finite-interval truncated normal (truncated_normal.py:68-82)"
    seen = np.zeros(4, dtype=int)
    for c in golden_factors["pwl3"]:
        ch = _pwl3(T, c, np.float64, ctx)
        got = ch.branch_masks(c["az"], c["bz"], c["ax"], c["bx"])
        with np.errstate(all="ignore"):
            exp = orc.pwl3_mid_masks(c["kind"], c["az"], c["bz"], c["ax"], c["bx"], c["param"])
        for k in range(4):
            assert np.array_equal(got[k], exp[k]), (c["kind"], k)
            seen[k] += int(exp[k].sum())
    assert seen[1] and seen[2] and seen[3]


def test_pwl_branch_masks_bit_exact(T, ctx, golden_factors):
    """log_Phi 'x < 30' masks.  (1) Against the mask the REFERENCE produced (tools/make_golden.py: log_Phi on
    x = ..., 29.9, nextafter(30, 0), 30, nextafter(30, 31), ...): with az = 1 the sign channel's pos region has
    r0 = bz, v0 = 1, so its log_Phi argument is bz itself, bit for bit, and the neg region's is -bz.  (2) On the
    golden channel cases, against the oracle's restatement of the same operand chain."""
    g = golden_factors["log_Phi"]
    xs, ref_mask = np.asarray(g["x"], dtype=np.float64), np.asarray(g["mask"], dtype=bool)
    assert ref_mask.tolist() == (xs < 30).tolist() and ref_mask.sum() == 10           # the fixture itself
    sgn = T.SgnChannel(ctx=ctx)
    m_pos, _ = sgn.branch_masks(1.0, xs, 0.7, np.zeros_like(xs))
    _, m_neg = sgn.branch_masks(1.0, -xs, 0.7, np.zeros_like(xs))
    assert np.array_equal(m_pos, ref_mask) and np.array_equal(m_neg, ref_mask)
    cls = dict(sgn=T.SgnChannel, abs=T.AbsChannel, relu=T.ReluChannel)
    for c in golden_factors["pwl"]:
        ch = cls[c["kind"]](ctx=ctx)
        with np.errstate(all="ignore"):
            m_pos, m_neg = ch.branch_masks(c["az"], c["bz"], c["ax"], c["bx"])
            exp = orc.pwl_log_Phi_masks(c["kind"], c["az"], c["bz"], c["ax"], c["bx"])
        assert np.array_equal(m_pos, exp[0])
        assert np.array_equal(m_neg, exp[1])


def _W(case):
    if "W" in case:
        return case["W"]
    rs = np.random.RandomState(case["seed"])
    return rs.randn(case["M"], case["N"]) / np.sqrt(case["N"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_linear_channel_golden(T, ctx, golden_linear, dtype):
    for case in golden_linear:
        W = _W(case)
        ch = T.LinearChannel(W, dtype=dtype, ctx=ctx, svd="host")
        assert ch.rank == case["rank"]
        bz, bx = case["bz"], case["bx"]
        for c in case["cases"]:
            az, ax = c["az"], c["ax"]
            if dtype == np.float32 and (az < 1e-3 or az > 1e6):
                continue
            cancel = 4e-16 * np.abs(bz).max() / max(az, 1e-11) * case["N"]
            rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
            rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
            close(rz, c["rz"], dtype, scale=np.abs(c["rz"]).max() * 10)
            close(rx, c["rx"], dtype, scale=np.abs(c["rx"]).max() * 10 + cancel / tol(dtype)["atol_scale"])
            # vz = (1 - n_eff)/az cancels when n_eff -> 1 (full column rank, az << ax)
            vz_atol = 64 * EPS[np.float64] / max(az, 1e-11)
            close(vz, c["vz"], atol=vz_atol); close(vx, c["vx"])
            v, a_new, n_eff = ch.device_variance(T._lib.FWD, az, ax)
            close(n_eff, c["n_eff"]); close(v, c["vx"])
            if "a_fwd" in c:
                a, b = ch.compute_forward_message(az, bz, ax, bx)
                close_a(a, c["a_fwd"], ax)
                close(b, c["b_fwd"], dtype,
                      scale=(np.abs(c["b_fwd"]).max() + np.abs(bx).max()) * 10
                      + cancel * (ax + c["a_fwd"]) / tol(dtype)["atol_scale"])
                a, b = ch.compute_backward_message(az, bz, ax, bx)
                close(a, c["a_bwd"], atol=(abs(az) + abs(c["a_bwd"])) * 64 * EPS[np.float64]
                      + vz_atol / c["vz"]**2)
                close(b, c["b_bwd"], dtype, scale=(np.abs(c["b_bwd"]).max() + np.abs(bz).max()) * 10,
                      atol=np.abs(c["rz"]).max() * vz_atol / c["vz"]**2)
        if "vz_az0" in case:
            v, _, _ = ch.device_variance(T._lib.BWD, 0.0, 0.7)
            close(v, case["vz_az0"])


@pytest.mark.parametrize("batch", [1, 3, 8, 11])
def test_batch_matches_per_instance_oracle(T, ctx, batch):
    """Ensemble axis: instance b of a batched call == the single-instance oracle on row b."""
    rs = np.random.RandomState(7)
    N, M = 96, 40
    W = rs.randn(M, N) / np.sqrt(N)
    lin = orc.LinearThinOracle(W)
    ch = T.LinearChannel(W, ctx=ctx, svd="host")
    az = rs.uniform(0.5, 3, batch); ax = rs.uniform(0.2, 5, batch)
    bz = rs.standard_normal((batch, N)); bx = rs.standard_normal((batch, M))
    af, bf = ch.compute_forward_message(az, bz, ax, bx)
    ab, bb = ch.compute_backward_message(az, bz, ax, bx)
    prior = T.GaussBernoulliPrior(size=N, rho=0.2, ctx=ctx)
    ap, bp = prior.compute_forward_message(az, bz)
    relu = T.ReluChannel(ctx=ctx)
    bx2 = rs.standard_normal((batch, N))
    ar, br = relu.compute_backward_message(az, bz, ax, bx2)
    for b in range(batch):
        a, bb_ = lin.forward_message(az[b], bz[b], ax[b], bx[b])
        close(af[b], a); close(bf[b], bb_, scale=10 * np.abs(bb_).max())
        a, bb_ = lin.backward_message(az[b], bz[b], ax[b], bx[b])
        close(ab[b], a); close(bb[b], bb_, scale=10 * np.abs(bb_).max())
        a, bb_ = orc.gb_prior_message(az[b], bz[b], rho=0.2)
        close(ap[b], a); close(bp[b], bb_, scale=10 * np.abs(bb_).max())
        a, bb_ = orc.pwl_channel_message("relu", "bwd", az[b], bz[b], ax[b], bx2[b])
        close(ar[b], a); close(br[b], bb_, scale=10 * np.abs(bb_).max())


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1023, 4099, 8191, 70001, 70002])
def test_ragged_and_empty_sizes(T, ctx, n):
    rs = np.random.RandomState(n)
    b = rs.standard_normal(n) * 2
    p = T.GaussBernoulliPrior(size=n, rho=0.3, ctx=ctx)
    if n == 0:
        a_new, b_new = p.compute_forward_message(1.1, b.reshape(1, 0))
        assert b_new.shape == (1, 0)
        return
    a_new, b_new = p.compute_forward_message(1.1, b)
    a_ref, b_ref = orc.gb_prior_message(1.1, b, rho=0.3)
    close(a_new, a_ref); close(b_new, b_ref, scale=10 * np.abs(b_ref).max())
    y = np.sign(rs.standard_normal(n))
    lk = T.SgnLikelihood(y=y, ctx=ctx)
    a_new, b_new = lk.compute_backward_message(0.9, b)
    a_ref, b_ref = orc.sgn_lik_message(0.9, b, y)
    close(a_new, a_ref); close(b_new, b_ref, scale=10 * np.abs(b_ref).max())
    # the two-region channels and the abs likelihood on the same ragged length, three instances (tile tails with
    # part of a wave inactive, the non-vector kernels for odd n)
    B = 3
    bz = rs.standard_normal((B, n)) * 1.5; bx = rs.standard_normal((B, n)) * 1.5
    az = np.array([1.3, 0.6, 2.2]); ax = np.array([0.7, 1.9, 0.4])
    for kind in ("relu", "sgn", "abs"):
        ch = {"relu": T.ReluChannel, "sgn": T.SgnChannel, "abs": T.AbsChannel}[kind](ctx=ctx)
        for direction in ("fwd", "bwd"):
            fn = ch.compute_forward_message if direction == "fwd" else ch.compute_backward_message
            a_dev, b_dev = fn(az, bz, ax, bx)
            for i in range(B):
                a_ref, b_ref = orc.pwl_channel_message(kind, direction, az[i], bz[i], ax[i], bx[i])
                close(a_dev[i], a_ref); close(b_dev[i], b_ref, scale=10 * np.abs(b_ref).max())
    ya = np.abs(rs.standard_normal((B, n)))
    a_dev, b_dev = T.AbsLikelihood(y=ya, ctx=ctx).compute_backward_message(az, bz)
    for i in range(B):
        a_ref, b_ref = orc.abs_lik_message(az[i], bz[i], ya[i])
        close(a_dev[i], a_ref); close(b_dev[i], b_ref, scale=10 * np.abs(b_ref).max())


def test_damping_fused(T, ctx):
    rs = np.random.RandomState(3)
    n, d = 1000, 0.3
    b = rs.standard_normal(n)
    old_b = rs.standard_normal(n); old_a = 0.7
    p = T.GaussBernoulliPrior(size=n, rho=0.3, ctx=ctx)
    oa = ctx.scalars(old_a); ob = ctx.array(old_b.reshape(1, n))
    p.compute_forward_message(ctx.scalars(1.4), ctx.array(b.reshape(1, n)), out_a=oa, out_b=ob,
                              damping=d, old_a=oa, old_b=ob)
    a_ref, b_ref = orc.gb_prior_message(1.4, b, rho=0.3)
    close(oa.get()[0], d * old_a + (1 - d) * a_ref)
    close(ob.get()[0], d * old_b + (1 - d) * b_ref, scale=10)


def test_nan_poisons_precision(T, ctx):
    b = np.array([1.0, np.nan, 2.0, 0.5])
    p = T.GaussBernoulliPrior(size=4, rho=0.3, ctx=ctx)
    a_new, _ = p.compute_forward_message(1.0, b)
    assert np.isnan(a_new)


def test_error_conventions(T, ctx):
    p = T.GaussBernoulliPrior(size=4, rho=0.3, ctx=ctx)
    with pytest.raises(ValueError):
        p.compute_forward_message(1.0, np.zeros((2, 2, 2)))
    with pytest.raises(T._lib.TrampHipError):
        T.GaussBernoulliPrior(size=4, rho=1.5, ctx=ctx).compute_forward_message(1.0, np.zeros(4))
    ch = T.LinearChannel(np.eye(4), ctx=ctx, svd="host")
    with pytest.raises(ValueError):
        ch.compute_forward_message(1.0, np.zeros(5), 1.0, np.zeros(4))
    with pytest.raises(AssertionError):
        T.SgnLikelihood(y=None, ctx=ctx).compute_backward_error(0.5, 1.0)   # az <= 1/tau_z


def test_large_properties_full_size(T, ctx):
    """BASELINE sizes (N = 1e6): size-independent properties instead of a CPU loop."""
    N = 1_000_000
    for dtype in (np.float64, np.float32):
        bx = ctx.randn((1, N), dtype, seed=5, scale=1.5)
        ax = ctx.scalars(1.3)
        p = T.GaussBernoulliPrior(size=N, rho=0.05, dtype=dtype, ctx=ctx)
        a1, b1 = p.compute_forward_message(ax, bx)
        a2, b2 = p.compute_forward_message(ax, bx)
        assert np.array_equal(a1.get(), a2.get()) and np.array_equal(b1.get(), b2.get())   # deterministic
        # consistency with the oracle on the SAME device-generated data
        bh = bx.get()[0].astype(np.float64)
        a_ref, b_ref = orc.gb_prior_message(1.3, bh, rho=0.05)
        close(a1.get()[0], a_ref, dtype)
        close(b1.get()[0], b_ref, dtype, scale=10 * np.abs(b_ref).max())
        # symmetry: flipping the sign of b flips r (zero-mean prior) and keeps v
        neg = ctx.array(-bx.get(), dtype)
        a3, b3 = p.compute_forward_message(ax, neg)
        close(a3.get(), a1.get(), dtype)
        close(b3.get(), -b1.get(), dtype, scale=10 * np.abs(b_ref).max())
        # L1: count equals the number of |b| > gamma exactly
        l1 = T.MAP_L1NormPrior(size=N, gamma=1.0, dtype=dtype, ctx=ctx)
        mask, count = l1.support(ax, bx)
        hb = bx.get()[0]
        assert int(count[0]) == int((np.abs(hb) > hb.dtype.type(1.0)).sum()) == int(mask.sum())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_large_properties_likelihoods_full_size(T, ctx, dtype):
    """BASELINE vector size N = 1e6 for the sign / abs output denoisers (K4, K5): determinism, the
    oracle on the same device-generated data, and the sign symmetry r(-b, -y) = -r(b, y) (sgn),
    r(-b, y) = -r(b, y) (abs).  tramp/likelihoods/sgn_likelihood.py:32-37, abs_likelihood.py:31-36."""
    N = 1_000_000
    bz = ctx.randn((1, N), dtype, seed=11, scale=1.2)
    yh = np.sign(ctx.randn((1, N), dtype, seed=12).get()[0].astype(np.float64))
    yh[yh == 0] = 1.0
    az = ctx.scalars(1.3)
    bh = bz.get()[0].astype(np.float64)
    for name in ("sgn", "abs"):
        y = yh if name == "sgn" else np.abs(ctx.randn((1, N), dtype, seed=13).get()[0].astype(np.float64)).astype(dtype).astype(np.float64)
        lk = (T.SgnLikelihood if name == "sgn" else T.AbsLikelihood)(y=y, dtype=dtype, ctx=ctx)
        a1, b1 = lk.compute_backward_message(az, bz)
        a2, b2 = lk.compute_backward_message(az, bz)
        assert np.array_equal(a1.get(), a2.get()) and np.array_equal(b1.get(), b2.get())
        ref = (orc.sgn_lik_message if name == "sgn" else orc.abs_lik_message)(1.3, bh, y)
        close_a(a1.get()[0], ref[0], 1.3, dtype)
        close(b1.get()[0], ref[1], dtype, scale=10 * np.abs(ref[1]).max())
        neg = ctx.array(-bz.get(), dtype)
        lk2 = (T.SgnLikelihood(y=-y, dtype=dtype, ctx=ctx) if name == "sgn" else lk)
        a3, b3 = lk2.compute_backward_message(az, neg)
        close(a3.get(), a1.get(), dtype)
        close(b3.get(), -b1.get(), dtype, scale=10 * np.abs(ref[1]).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["sgn", "abs", "relu"])
def test_large_properties_pwl_channels_full_size(T, ctx, dtype, kind):
    """N = 1e6 for the Sgn / Abs / Relu channels (K7), both directions: determinism, the oracle on the
    same data, branch masks bit-exact in f64, and the symmetries of the odd (sgn) / even (abs) channel
    under z -> -z.  tramp/channels/activation/piecewise_linear_channel.py:27-53,80-106."""
    N = 1_000_000
    bz = ctx.randn((1, N), dtype, seed=21, scale=1.1)
    bx = ctx.randn((1, N), dtype, seed=22, scale=0.9)
    az, ax = ctx.scalars(1.3), ctx.scalars(0.7)
    ch = {"sgn": T.SgnChannel, "abs": T.AbsChannel, "relu": T.ReluChannel}[kind](dtype=dtype, ctx=ctx)
    bzh, bxh = bz.get()[0].astype(np.float64), bx.get()[0].astype(np.float64)
    for d in ("fwd", "bwd"):
        call = ch.compute_forward_message if d == "fwd" else ch.compute_backward_message
        a1, b1 = call(az, bz, ax, bx)
        a2, b2 = call(az, bz, ax, bx)
        assert np.array_equal(a1.get(), a2.get()) and np.array_equal(b1.get(), b2.get())
        a_ref, b_ref = orc.pwl_channel_message(kind, d, 1.3, bzh, 0.7, bxh)
        a_in = 0.7 if d == "fwd" else 1.3
        close_a(a1.get()[0], a_ref, a_in, dtype)
        close(b1.get()[0], b_ref, dtype, scale=10 * np.abs(b_ref).max())
        if dtype == np.float64 and d == "bwd":
            # the masks at full size, on device-generated messages stretched so that both sides of the threshold occur
            wide = ctx.array(bz.get() * 40.0, dtype)
            m_pos, m_neg = ch.branch_masks(az, wide, ax, bx)
            exp = orc.pwl_log_Phi_masks(kind, 1.3, wide.get()[0], 0.7, bxh)
            assert np.array_equal(m_pos[0] if m_pos.ndim == 2 else m_pos, exp[0])
            assert np.array_equal(m_neg[0] if m_neg.ndim == 2 else m_neg, exp[1])
            assert 0 < exp[0].sum() < N and 0 < exp[1].sum() < N
        if kind in ("sgn", "abs"):
            # z -> -z: sgn is odd (x -> -x as well), abs is even (x unchanged)
            nz = ctx.array(-bz.get(), dtype)
            nx = ctx.array(-bx.get(), dtype) if kind == "sgn" else bx
            a3, b3 = call(az, nz, ax, nx)
            close(a3.get(), a1.get(), dtype)
            sign = -1.0 if (kind == "sgn" or d == "bwd") else 1.0
            close(b3.get(), sign * b1.get(), dtype, scale=10 * np.abs(b_ref).max())


@pytest.mark.parametrize("kind", ["sgn", "abs", "relu"])
def test_pwl_channels_non_finite_messages_give_nan(T, ctx, kind):
    """A NaN or infinite component of either incoming message makes the reference's posterior NaN at that component, in
    every channel and both directions (inf - inf in the region log-partitions, linear_region.py:55-64; checked on the
    reference itself when this test was written), and hence a_new NaN.  Sgn forward used to return a finite r there (its
    regions have slope 0 and never look at the truncated mean)."""
    cls = dict(sgn=T.SgnChannel, abs=T.AbsChannel, relu=T.ReluChannel)
    ch = cls[kind](ctx=ctx)
    for which in ("bz", "bx"):
        for val in (np.nan, np.inf, -np.inf):
            bz = np.array([0.3, 0.5, -0.2, 1.0, -1.5, 0.7]); bx = np.array([0.1, 0.2, 0.3, -0.4, 0.5, -0.6])
            (bz if which == "bz" else bx)[1] = val
            for post, msg in ((ch.compute_forward_posterior, ch.compute_forward_message),
                              (ch.compute_backward_posterior, ch.compute_backward_message)):
                r, v = post(1.3, bz, 0.7, bx)
                r = np.ravel(r)
                assert np.isnan(r[1]) and np.all(np.isfinite(np.delete(r, 1))) and np.isnan(v), (kind, which, val, r, v)
                a_new, _ = msg(1.3, bz, 0.7, bx)
                assert np.isnan(a_new)


def test_linear_channel_sample_with_and_without_a_kept_matrix(ctx):
    "LinearChannel.sample (linear_channel.py:48-50): a host W stays referenced; from a device W the factors reproduce W z"
    import tramp_amd as T
    rs = np.random.RandomState(3)
    W = rs.standard_normal((40, 70)) / np.sqrt(70)
    z = rs.standard_normal(70)
    ch = T.LinearChannel(W, ctx=ctx)
    assert ch.W is W
    np.testing.assert_allclose(ch.sample(z), W @ z, rtol=1e-14)
    chd = T.LinearChannel(ctx.array(W, np.float64), ctx=ctx, svd="device")
    assert chd.W is None
    np.testing.assert_allclose(chd.sample(z), W @ z, rtol=1e-11, atol=1e-12)
    Z = rs.standard_normal((3, 70))
    np.testing.assert_allclose(chd.sample(Z), Z @ W.T, rtol=1e-11, atol=1e-12)
