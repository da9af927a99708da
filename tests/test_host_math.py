"""CPU unit tests of the kernels' arithmetic: tramp_amd/csrc/tramp_math.h compiled for the host
(tests/host_math, test infrastructure only) against the oracle and the reference goldens.
Covers both the literal restatement and the hoisted 'fast' forms the streaming kernels use."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import tramp_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
KIND = dict(sgn=0, abs=1, relu=2)
EPS64 = float(np.finfo(np.float64).eps)


@pytest.fixture(scope="module")
def hm():
    d = os.path.join(HERE, "host_math")
    subprocess.run(["make", "-C", d], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(d, "libhostmath.so"))
    lib.hm_a_new.restype = C.c_double
    lib.hm_a_new.argtypes = [C.c_double, C.c_double]
    return lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def close(a, b, rtol=1e-11, scale=None):
    b = np.asarray(b, dtype=float)
    sc = np.nanmax(np.abs(b)) if scale is None else scale
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(sc, 1e-300), equal_nan=True)


@pytest.mark.parametrize("fast", [0, 1])
def test_gb(hm, golden_factors, fast):
    for c in golden_factors["gb"]:
        b = np.ascontiguousarray(c["b"], dtype=np.float64)
        r, v = np.empty_like(b), np.empty_like(b)
        hm.hm_gb_f64(C.c_size_t(b.size), ptr(b), C.c_double(c["ax"]), C.c_double(c["rho"]), C.c_double(c["mean"]),
                     C.c_double(c["var"]), ptr(r), ptr(v), fast)
        close(r, c["r"]); close(v, c["v_diag"])
        b32 = b.astype(np.float32)
        if np.all(np.isfinite(b32)) and np.abs(b).max() < 1e4 and 1e-6 < c["ax"] < 1e6:
            r32, v32 = np.empty_like(b32), np.empty_like(b32)
            hm.hm_gb_f32(C.c_size_t(b.size), ptr(b32), C.c_double(c["ax"]), C.c_double(c["rho"]),
                         C.c_double(c["mean"]), C.c_double(c["var"]), ptr(r32), ptr(v32), fast)
            close(r32, c["r"], rtol=2e-4); close(v32, c["v_diag"], rtol=2e-4)


def test_l1_and_mask(hm, golden_factors):
    for c in golden_factors["l1"]:
        b = np.ascontiguousarray(c["b"], dtype=np.float64)
        r, v, m = np.empty_like(b), np.empty_like(b), np.empty(b.size, dtype=np.uint8)
        hm.hm_l1_f64(C.c_size_t(b.size), ptr(b), C.c_double(c["ax"]), C.c_double(c["gamma"]), ptr(r), ptr(v), ptr(m))
        close(r, c["r"], rtol=1e-14)
        assert np.array_equal(m.astype(bool), c["mask"])            # bit-exact
        close(v.mean(), c["v"], rtol=1e-12)


@pytest.mark.parametrize("fast", [0, 1])
def test_sgn_abs_lik(hm, golden_factors, fast):
    for c in golden_factors["sgn_lik"]:
        bz, y = np.ascontiguousarray(c["bz"], dtype=float), np.ascontiguousarray(c["y"], dtype=float)
        r, v = np.empty_like(bz), np.empty_like(bz)
        hm.hm_sgn_lik_f64(C.c_size_t(bz.size), ptr(bz), ptr(y), C.c_double(c["az"]), ptr(r), ptr(v), fast)
        # v = v0 (1 + g2 - g1^2) cancels ~x^2-fold deep in the tail: the literal form's host stand-in
        # exp(x^2) erfc(x) for erfcx carries ~x^2 eps, so 1e-7 there; the fast form evaluates the kernels' own
        # table erfcx (3e-15) and holds what the GPU test holds
        close(r, c["r"], rtol=1e-9); close(v, c["v_diag"], rtol=1e-9 if fast else 1e-7)
    for c in golden_factors["abs_lik"]:
        bz, y = np.ascontiguousarray(c["bz"], dtype=float), np.ascontiguousarray(c["y"], dtype=float)
        r, v = np.empty_like(bz), np.empty_like(bz)
        hm.hm_abs_lik_f64(C.c_size_t(bz.size), ptr(bz), ptr(y), ptr(r), ptr(v))
        close(r, c["r"]); close(v, c["v_diag"])


@pytest.mark.parametrize("fast", [0, 1])
def test_pwl_and_masks(hm, golden_factors, fast):
    for c in golden_factors["pwl"]:
        if np.isnan(c["v_fwd"]) or np.isnan(c["v_bwd"]):
            continue
        bz, bx = np.ascontiguousarray(c["bz"], dtype=float), np.ascontiguousarray(c["bx"], dtype=float)
        for d, fwd in (("fwd", 1), ("bwd", 0)):
            r, v, m = np.empty_like(bz), np.empty_like(bz), np.empty(bz.size, dtype=np.uint8)
            hm.hm_pwl_f64(KIND[c["kind"]], fwd, C.c_size_t(bz.size), ptr(bz), ptr(bx), C.c_double(c["az"]),
                          C.c_double(c["ax"]), ptr(r), ptr(v), ptr(m), fast)
            with np.errstate(all="ignore"):
                r_ref, v_ref = orc.pwl_channel_posterior(c["kind"], d, c["az"], bz, c["ax"], bx, isotropic=False)
                rk = [orc.pwl_region_terms(sd, sl, x0, c["az"], bz, c["ax"], bx)[0] * (sl if fwd else 1)
                      + (x0 if fwd else 0) for sd, sl, x0 in orc.PWL_REGIONS[c["kind"]]]
            sc = max(1.0, np.nanmax(np.abs(rk)))
            close(r, r_ref, rtol=1e-10, scale=sc)
            close(v, v_ref, rtol=1e-9, scale=max(np.nanmax(np.abs(v_ref)), sc * sc * 1e-3))
            close(v.mean(), c["v_" + d], rtol=1e-9)
            # branch masks: same comparison on the same operands as the reference
            with np.errstate(all="ignore"):
                exp = 0
                for k, (side, slope, x0) in enumerate(orc.PWL_REGIONS[c["kind"]]):
                    a = c["az"] + slope**2 * c["ax"]
                    b = bz + slope * (bx - c["ax"] * x0)
                    y0 = (0 - b / a) / np.sqrt(1 / a)
                    exp = exp | (((-side * y0) < 30).astype(np.uint8) << k)
            assert np.array_equal(m, exp)


def pwl_truth_mp(kind, fwd, az, bz, ax, bx):
    """Posterior (r, v) of one component of a Sgn / Abs / Relu channel at 50 digits: the reference's formulas
    (piecewise_linear_channel.py:27-53, linear_region.py:32-64, truncated_normal.py:218-266) in mpmath, where the
    cancellations of their f64 evaluation do not exist."""
    import mpmath as mp
    mp.mp.dps = 50
    az, bz, ax, bx = (mp.mpf(float(t)) for t in (az, bz, ax, bx))
    logw, m1, m2 = [], [], []
    for sd, sl, x0 in orc.PWL_REGIONS[kind]:
        a = az + sl * sl * ax
        b = bz + sl * (bx - ax * x0)
        v0 = 1 / a
        r0, s0 = b * v0, mp.sqrt(v0)
        x = sd * (0 - r0) / (s0 * mp.sqrt(2))
        ec = mp.erfc(x)
        g = mp.exp(-x * x) / ec
        mean = r0 + sd * s0 * mp.sqrt(2 / mp.pi) * g
        var = v0 * (1 + 2 * x * g / mp.sqrt(mp.pi) - (2 / mp.pi) * g * g)
        logw.append(mp.log(s0) + mp.log(ec) + r0 * r0 / (2 * v0) - ax * x0 * x0 / 2 + bx * x0)
        mk = sl * mean + x0 if fwd else mean
        vk = sl * sl * var if fwd else var
        m1.append(mk)
        m2.append(vk + mk * mk)
    mx = max(logw)
    w = [mp.exp(t - mx) for t in logw]
    S = sum(w)
    r = sum(wi * mi for wi, mi in zip(w, m1)) / S
    v = sum(wi * mi for wi, mi in zip(w, m2)) / S - r * r
    return float(r), float(v)


@pytest.mark.parametrize("kind", ["sgn", "abs", "relu"])
@pytest.mark.parametrize("fast", [1, 2])
def test_pwl_moment_form_dense(hm, kind, fast):
    """The moment form of K7 (tramp_math.h: pwl_moment_form, one reciprocal per component) against the literal region form
    of the reference (piecewise_linear_channel.py:27-53) on dense random messages: typical Bayes-optimal scales, wide ones
    that cross the edges of the erfcx grid and the log_Phi mask threshold, strongly offset ones (|r| >> sqrt v), extreme
    precisions; both table variants (1 = uniform grid, 2 = the small table of the resident pipeline).  Masks bit for bit;
    means and the instance average of v as the GPU tests ask; per component, against a 50-digit evaluation on a sample,
    the new form is at least as accurate as the reference's own f64 arithmetic (whose var_k = v0 (1 + g2 - g1^2) cancels
    in the tails)."""
    rs = np.random.RandomState(5)
    n = 20000
    for az, ax, sz, sx, off in ((1.3, 0.7, 1.5, 1.5, 0.0), (1.3, 0.7, 6.0, 6.0, 0.0), (2.0, 1e-6, 40.0, 1.0, 0.0),
                                (1e-3, 5.0, 0.2, 30.0, 0.0), (50.0, 80.0, 100.0, 150.0, 0.0), (1.0, 1.0, 1.0, 1.0, 300.0),
                                (1.0, 2.0, 1.0, 1.0, -50.0), (1e4, 1e-4, 3e2, 1e-2, 0.0), (3.0, 0.0, 2.0, 0.0, 0.0)):
        bz = (off + sz * rs.standard_normal(n)).astype(np.float64)
        bx = (sx * rs.standard_normal(n) - 0.5 * off).astype(np.float64)
        for fwd in (0, 1):
            if ax == 0.0 and fwd:
                continue
            out = {}
            for f in (0, fast):
                r, v, m = np.empty(n), np.empty(n), np.empty(n, dtype=np.uint8)
                hm.hm_pwl_any_f64(KIND[kind], fwd, C.c_size_t(n), ptr(bz), ptr(bx), C.c_double(az), C.c_double(ax), ptr(r), ptr(v),
                                  ptr(m), f)
                out[f] = (r, v, m)
            (r0, v0, m0), (r1, v1, m1) = out[0], out[fast]
            assert np.array_equal(m0, m1)
            ok = np.isfinite(r0) & np.isfinite(v0)            # the literal form overflows (erfcx) far under the mask
            assert ok.mean() > 0.5
            with np.errstate(all="ignore"):
                rk = [orc.pwl_region_terms(sd, sl, x0, az, bz, ax, bx)[0] * (sl if fwd else 1) + (x0 if fwd else 0)
                      for sd, sl, x0 in orc.PWL_REGIONS[kind]]
            sc = np.maximum(1.0, np.maximum(np.abs(rk[0]), np.abs(rk[1])))
            assert np.all(np.abs(r1 - r0)[ok] <= 1e-10 * sc[ok]), np.max((np.abs(r1 - r0) / sc)[ok])
            assert abs(v1[ok].mean() - v0[ok].mean()) <= 1e-10 * abs(v0[ok].mean()) + 1e-290       # (a capped p = 1e-300 stands for 0)
            # per component against 50 digits, on the components where the two forms differ most and on a random sample
            rel = np.where(ok, np.abs(v1 - v0) / np.maximum(np.abs(v0), 1e-300), 0.0)
            idx = np.unique(np.concatenate([np.argsort(rel)[-40:], rs.choice(n, 60, replace=False)]))
            for i in idx[ok[idx]]:
                rt, vt = pwl_truth_mp(kind, fwd, az, bz[i], ax, bx[i])
                if not (np.isfinite(rt) and np.isfinite(vt)) or vt <= 0:
                    continue
                e_lit, e_new = abs(v0[i] - vt), abs(v1[i] - vt)
                # which form this component took (the host harness votes per component): both erfcx arguments on the
                # grid [-7.5, 8) - or, Abs / Relu, one below it with the other >= -4 - (fast = 1) / below 0.98 x 30 / sqrt 2 in magnitude (fast = 2)
                xs = [-sd * (bz[i] + sl * (bx[i] - ax * x0)) / np.sqrt(2 * (az + sl * sl * ax)) for sd, sl, x0 in orc.PWL_REGIONS[kind]]
                lim = 0.98 * 30 / np.sqrt(2)
                moment = ((max(xs) < 8 and (min(xs) >= -7.5 or (kind != "sgn" and max(xs) >= -4 and min(xs) > -lim)))
                          if fast == 1 else max(abs(t) for t in xs) < lim)
                slack = 1e-13 if moment else 5e-11            # (the region-by-region form of rounds 2-4 reaches ~1e-11)
                # a region deep in its tail has mean = r0 + s0 g / e and second moment v0 + r0^2 - ... cancelling in ANY
                # f64 form, the reference's included: a few ulp of the largest v0_k + r0_k^2 on top
                cond = max(1 / (az + sl * sl * ax) + ((bz[i] + sl * (bx[i] - ax * x0)) / (az + sl * sl * ax)) ** 2
                           for sd, sl, x0 in orc.PWL_REGIONS[kind])
                assert e_new <= 4 * e_lit + slack * abs(vt) + 8 * EPS64 * cond, (az, ax, fwd, i, bz[i], bx[i], moment, e_lit / vt, e_new / vt)
                assert abs(r1[i] - rt) <= 4 * abs(r0[i] - rt) + 1e-13 * sc[i], (i, bz[i], bx[i])      # exp(x^2) carries x^2 ulp
        # f32: the same comparison at the north star's 1e-3
        if 1e-2 < az < 1e3 and (ax == 0 or 1e-2 < ax < 1e3) and sz < 50 and sx < 50:
            bzf, bxf = bz.astype(np.float32), bx.astype(np.float32)
            for fwd in (0, 1):
                if ax == 0.0 and fwd:
                    continue
                r, v, m = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, dtype=np.uint8)
                hm.hm_pwl_any_f32(KIND[kind], fwd, C.c_size_t(n), ptr(bzf), ptr(bxf), C.c_double(az), C.c_double(ax), ptr(r), ptr(v),
                                  ptr(m), fast)
                r64, v64, m64 = np.empty(n), np.empty(n), np.empty(n, dtype=np.uint8)
                hm.hm_pwl_any_f64(KIND[kind], fwd, C.c_size_t(n), ptr(bzf.astype(np.float64)), ptr(bxf.astype(np.float64)),
                                  C.c_double(az), C.c_double(ax), ptr(r64), ptr(v64), ptr(m64), 0)
                ok = np.isfinite(r64) & np.isfinite(v64) & np.isfinite(r)
                assert ok.mean() > 0.9
                sc = np.maximum(1.0, np.abs(r64))
                assert np.all(np.abs(r - r64)[ok] <= 1e-3 * sc[ok]), np.max((np.abs(r - r64) / sc)[ok])
                assert abs(v[ok].astype(np.float64).mean() - v64[ok].mean()) <= 1e-3 * abs(v64[ok].mean()) + 1e-30


def test_widened_factors(hm, golden_factors):
    "SURVEY section 8f rank 2 factors: generic two-half-line, three-region channels, (Leaky)Relu likelihoods"
    spec3 = {"h-tanh": ((-1., 1.), (0., 1., 0.), (1., 0., -1.)), "h-sigm": ((-2.5, 2.5), (0., .2, 0.), (1., .5, 0.))}
    for c in golden_factors["pwl3"]:
        (zlo, zhi), sl, x0 = spec3.get(c["kind"], ((-(c["param"] or 0), c["param"] or 0), (0., 0., 0.), (1., -1., 1.)))
        bz, bx = np.ascontiguousarray(c["bz"], dtype=float), np.ascontiguousarray(c["bx"], dtype=float)
        sl, x0 = np.array(sl), np.array(x0)
        for d, fwd in (("fwd", 1), ("bwd", 0)):
            r, v, m = np.empty_like(bz), np.empty_like(bz), np.empty(bz.size, dtype=np.uint8)
            hm.hm_pwl3_f64(fwd, C.c_size_t(bz.size), ptr(bz), ptr(bx), C.c_double(c["az"]), C.c_double(c["ax"]),
                           C.c_double(zlo), C.c_double(zhi), ptr(sl), ptr(x0), ptr(r), ptr(v), ptr(m))
            with np.errstate(all="ignore"):
                r_ref, v_ref = orc.pwl3_channel_posterior(c["kind"], d, c["az"], bz, c["ax"], bx, isotropic=False,
                                                          param=c["param"])
                exp = orc.pwl3_mid_masks(c["kind"], c["az"], bz, c["ax"], bx, c["param"])
            close(v.mean(), c["v_" + d], rtol=1e-8)
            close(r, r_ref, rtol=1e-8, scale=max(1.0, np.nanmax(np.abs(r_ref))))
            code = (m >> 2) & 3
            for k in range(4):
                assert np.array_equal(code == k, exp[k])
    for c in golden_factors["pwl2"]:
        spec = (1., c["param"], 0., 0., 0.) if c["kind"] == "l-relu" else (1., -1., 0., 0., c["param"])
        bz, bx = np.ascontiguousarray(c["bz"], dtype=float), np.ascontiguousarray(c["bx"], dtype=float)
        for d, fwd in (("fwd", 1), ("bwd", 0)):
            r, v = np.empty_like(bz), np.empty_like(bz)
            hm.hm_pwl2_f64(fwd, C.c_size_t(bz.size), ptr(bz), ptr(bx), C.c_double(c["az"]), C.c_double(c["ax"]),
                           *[C.c_double(x) for x in spec], ptr(r), ptr(v))
            close(v.mean(), c["v_" + d], rtol=1e-8)
    for c in golden_factors["relu_lik"]:
        bz, y = np.ascontiguousarray(c["bz"], dtype=float), np.ascontiguousarray(c["y"], dtype=float)
        r, v = np.empty_like(bz), np.empty_like(bz)
        hm.hm_relu_lik_f64(C.c_size_t(bz.size), ptr(bz), ptr(y), C.c_double(c["az"]), C.c_double(c["slope"]), ptr(r), ptr(v))
        close(r, c["r"], rtol=1e-9, scale=max(1.0, np.abs(c["r"]).max()))
        close(v.mean(), c["v"], rtol=1e-7, scale=max(float(c["v"]), 1e-300))


def test_a_new_clipping_and_nan(hm):
    assert hm.hm_a_new(0.5, 1.0) == orc.a_new(0.5, 1.0)
    assert hm.hm_a_new(1e-30, 0.0) == 1e11 and hm.hm_a_new(10.0, 5.0) == 1e-11
    assert np.isnan(hm.hm_a_new(float("nan"), 1.0))


def _special(hm, which, x, f32=False):
    x = np.ascontiguousarray(x, dtype=np.float32 if f32 else np.float64)
    out = np.empty_like(x)
    (hm.hm_special_f32 if f32 else hm.hm_special_f64)(which, C.c_size_t(x.size), ptr(x), ptr(out))
    return out


def test_special_functions_of_the_streaming_kernels(hm):
    """The DEVICE expressions of tramp_math.h (table erfcx on both grids, the polynomial exp, tanh, the corrected
    reciprocal, expit), compiled for the host with only the hardware primitives replaced (tmath::hw), against scipy at
    the accuracies their comments state."""
    from scipy import special
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-6, 30, 20000), rng.uniform(-7.5, 8, 20000), 10.0 ** rng.uniform(-8, 8, 2000),
                        [0.0, -0.0, 1e300, 4.0, 8.0 - 1e-12, -3.0, -3.0 - 1e-12, -5.0, -7.5, -7.5 - 1e-12, -7.5 + 1e-12, 26.5]])
    for which in (0, 1):
        got, exp = _special(hm, which, x), special.erfcx(x)
        rel = np.abs(got - exp) / exp
        if which == 1:
            # the one-value-per-point Taylor table of round 5 (erfcx_ttable.h): 2e-15 on [-3, 8), its degree-6 truncation
            # grows to 4e-14 at -5 and 6e-13 at -7.5, where erfcx = 1.4e11 / 5e24 is the weight of a region that then
            # holds all the mass
            assert np.max(rel[(x >= -3) | (x < -7.5)]) < 1e-14 and np.max(rel[x >= -5]) < 1e-13 and np.max(rel) < 1e-12
        else:
            assert np.max(rel) < 1e-14
    # contract (tramp_math.h): a NaN argument is swallowed by the clamp - the callers' own use of x carries it
    assert np.isfinite(_special(hm, 0, [np.nan])[0]) and np.isfinite(_special(hm, 1, [np.nan])[0])
    xe = np.concatenate([rng.uniform(-745, 709, 20000), rng.uniform(-2, 2, 5000), [0.0, -745.13, 709.78]])
    got, exp = _special(hm, 2, xe), np.exp(xe)
    ok = exp > 1e-300                                     # denormal results: one rounding of ldexp, not relative
    assert np.max(np.abs(got[ok] - exp[ok]) / exp[ok]) < 5e-16
    np.testing.assert_allclose(got[~ok], exp[~ok], rtol=2e-15, atol=2e-323)
    assert _special(hm, 2, [-800.0])[0] == 0.0 and np.isinf(_special(hm, 2, [800.0])[0]) and np.isnan(_special(hm, 2, [np.nan])[0])
    assert _special(hm, 3, [-800.0])[0] <= 5e-324 * 4 and np.isfinite(_special(hm, 3, [800.0])[0])
    xt = np.concatenate([rng.uniform(-30, 30, 20000), rng.uniform(-0.3, 0.3, 5000), [0.0, 0.25, -0.25, 1e-300]])
    got, exp = _special(hm, 4, xt), np.tanh(xt)
    assert np.max(np.abs(got - exp) / np.maximum(np.abs(exp), 1e-300)) < 2e-13
    # the table form (tanh_table.h, round 5): the binary prior's
    xt = np.concatenate([xt, 10.0 ** rng.uniform(-12, 0, 3000), (np.arange(1249) + 0.5) / 64, [19.5, 19.5 + 1e-9, -25.0, 700.0, np.inf, -np.inf]])
    got, exp = _special(hm, 7, xt), np.tanh(xt)
    assert np.max(np.abs(got - exp) / np.maximum(np.abs(exp), 1e-300)) < 1e-15
    assert np.isnan(_special(hm, 7, [np.nan])[0]) and _special(hm, 7, [-0.0])[0] == 0.0
    xr = 10.0 ** rng.uniform(-300, 300, 20000)
    assert np.max(np.abs(_special(hm, 5, xr) * xr - 1.0)) < 4e-16
    xs = rng.uniform(-800, 800, 20000)
    np.testing.assert_allclose(_special(hm, 6, xs), special.expit(xs), rtol=1e-15, atol=1e-300)
    # f32 forms
    x32 = np.concatenate([rng.uniform(0, 30, 5000), rng.uniform(-3, 0, 2000)]).astype(np.float32)
    got, exp = _special(hm, 0, x32, f32=True), special.erfcx(x32.astype(np.float64))
    assert np.max(np.abs(got - exp) / exp) < 3e-6
    xt32 = np.concatenate([rng.uniform(-12, 12, 5000), 10.0 ** rng.uniform(-8, 0, 1000), [9.5, 9.6, -40.0, np.inf]]).astype(np.float32)
    got, exp = _special(hm, 7, xt32, f32=True), np.tanh(xt32.astype(np.float64))
    assert np.max(np.abs(got - exp) / np.maximum(np.abs(exp), 1e-30)) < 4e-7 and np.isnan(_special(hm, 7, np.float32([np.nan]), f32=True)[0])
    xe32 = rng.uniform(-85, 88, 5000).astype(np.float32)           # normal f32 results
    got, exp = _special(hm, 3, xe32, f32=True), np.exp(xe32.astype(np.float64))
    assert np.max(np.abs(got - exp) / exp) < 1e-6


# ---------------------------------------------------------------- round 6: remaining priors, modulus likelihood
def test_halfline_and_mixture_priors(hm, golden_priors2):
    n_run = 0
    for c in golden_priors2["cases"]:
        if c.get("diag"):
            continue
        bx, ax, kw = np.ascontiguousarray(c["bx"]), float(c["ax"]), c["kwargs"]
        r, v = np.empty_like(bx), np.empty_like(bx)
        if c["name"] in ("exponential", "positive"):
            a0, b0 = (0.0, -1.0 / kw.get("mean", 1)) if c["name"] == "exponential" else (1.0, 0.0)
            hm.hm_halfline_prior_f64(C.c_size_t(bx.size), ptr(bx), C.c_double(ax), C.c_double(a0), C.c_double(b0), ptr(r), ptr(v))
        else:
            K = len(kw["probs"])
            pr, me, va = (np.asarray(kw[k], dtype=np.float64) for k in ("probs", "means", "vars"))
            # the run-time form (K <= 8) and the straight-line form of the streaming kernels (K = 2..4, polynomial exp)
            hm.hm_gm_f64(C.c_size_t(bx.size), ptr(bx), C.c_double(ax), C.c_int(K), ptr(pr), ptr(me), ptr(va), ptr(r), ptr(v), 0)
            close(r, c["r"], 1e-11)
            close(v.mean(), c["v"], 1e-10)
            hm.hm_gm_f64(C.c_size_t(bx.size), ptr(bx), C.c_double(ax), C.c_int(K), ptr(pr), ptr(me), ptr(va), ptr(r), ptr(v), 1)
            b32 = bx.astype(np.float32)
            if np.abs(bx).max() * max(1.0, 1 / ax) < 1e3:          # f32 form inside its range
                r32, v32 = np.empty_like(b32), np.empty_like(b32)
                hm.hm_gm_f32(C.c_size_t(bx.size), ptr(b32), C.c_double(ax), C.c_int(K), ptr(pr), ptr(me), ptr(va), ptr(r32), ptr(v32), 1)
                close(r32, c["r"], 3e-4)
                close(v32.mean(dtype=np.float64), c["v"], 1e-3)
        close(r, c["r"], 1e-11)
        # the reference's v0 (1 + g2 - g1^2) cancels to eps x^2 in the far tail (b = -300 at a = 1e-3: x = 6.7e3, where it
        # returns 0.0 for a variance of 1e-5): its own round-off is 3e-9 of the mean there
        close(v.mean(), c["v"], 1e-8)
        n_run += 1
    assert n_run == 35


def test_bessel_ratio_and_modulus_posterior(hm, golden_complex):
    import mpmath as mp
    mp.mp.dps = 40
    x = np.concatenate([[0.0, 1e-300, 1e-8, 1e-3], np.linspace(0.01, 60.0, 400), [24.999, 25.0, 25.001, 1e3, 1e6, 9.9e8]])
    out = np.empty_like(x)
    hm.hm_bessel_ratio(C.c_size_t(x.size), ptr(x), ptr(out))
    ref = np.array([float(mp.besseli(1, mp.mpf(t)) / mp.besseli(0, mp.mpf(t))) for t in x])
    np.testing.assert_allclose(out, ref, rtol=4e-15, atol=1e-305)
    np.testing.assert_allclose(out, orc.ive_ratio(x), rtol=2e-14, atol=1e-305)
    edge = np.array([1e9, 2e9, np.inf, np.nan, -3.0])
    got = np.empty_like(edge)
    hm.hm_bessel_ratio(C.c_size_t(edge.size), ptr(edge), ptr(got))
    assert list(got[:4]) == [1.0, 1.0, 1.0, 1.0] and abs(got[4] + float(mp.besseli(1, 3) / mp.besseli(0, 3))) < 1e-15
    for c in golden_complex["modulus"]:
        re, im, y = (np.ascontiguousarray(a) for a in (c["bz"][0], c["bz"][1], c["y"]))
        rre, rim, v = np.empty_like(re), np.empty_like(re), np.empty_like(re)
        hm.hm_modulus(C.c_size_t(re.size), ptr(re), ptr(im), ptr(y), ptr(rre), ptr(rim), ptr(v))
        close(np.stack([rre, rim]), c["r"], 1e-13)
        close(v.mean(), c["v"], 1e-12)


def test_mixture_narrow_component_far_from_zero_keeps_its_digits_in_f32(hm):
    """a component with mean^2 / var = 4e4 (mean 2, var 1e-4): eta_k and normal.A(a_k, b_k) are each ~2e4 and cancel - the
    kernels' constants take them against each other analytically (tramp_math.h: gm_const_k), so f32 messages stay at 1e-5"""
    rs = np.random.RandomState(3)
    pr, me, va = (np.array(t, dtype=np.float64) for t in ([0.3, 0.7], [2.0, -0.5], [1e-4, 0.5]))
    for ax in (0.05, 1.3, 40.0):
        bx = (ax * np.where(rs.rand(400) < 0.3, 2.0, -0.5) + np.sqrt(ax) * rs.standard_normal(400)).astype(np.float32)
        r_o, v_o = orc.gm_prior_posterior(ax, bx.astype(np.float64), pr, me, va, isotropic=False)
        for fast in (0, 1):
            r32, v32 = np.empty_like(bx), np.empty_like(bx)
            hm.hm_gm_f32(C.c_size_t(bx.size), ptr(bx), C.c_double(ax), C.c_int(2), ptr(pr), ptr(me), ptr(va), ptr(r32), ptr(v32), fast)
            close(r32, r_o, 2e-5)
            close(v32, v_o, 2e-4)
