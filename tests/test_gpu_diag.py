"""GPU parity of the diagonal-belief (``isotropic=False``) variants of every separable factor built here, on the
inputs of the reference's own factor tests (tramp/tests/test_priors.py:17-51, tramp/tests/test_likelihoods.py:22-83:
a = linspace(1, 2, 100), b = linspace(-2, 2, 100), y = f(linspace(-3, 3, 100))) plus ragged random and clip-edge
inputs: posteriors (r, v arrays), messages (a_new, b_new arrays, element-wise clip of tramp/base.py:250-255) and
the mean log-partition, against tests/golden/diag.npz (captured from the reference by tools/make_golden_diag.py)
and against the oracle on the same inputs.  f64 1e-9, f32 1e-3."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc
from tests.conftest import diag_oracle_posterior

pytestmark = pytest.mark.gpu
EPS = {np.float64: 2.3e-16, np.float32: 1.2e-7}
RTOL = {np.float64: 1e-9, np.float32: 1e-3}


def build(T, ctx, c, dtype):
    kw = dict(c["kwargs"])
    n = c["b"].shape[0]
    if c["family"] == "prior":
        cls = {"gaussian": T.GaussianPrior, "gauss_bernoulli": T.GaussBernoulliPrior, "binary": T.BinaryPrior,
               "l1": T.MAP_L1NormPrior}[c["name"]]
        return cls(size=n, isotropic=False, dtype=dtype, ctx=ctx, **kw)
    cls = {"gaussian": T.GaussianLikelihood, "abs": T.AbsLikelihood, "sgn": T.SgnLikelihood, "relu": T.ReluLikelihood,
           "l-relu": T.LeakyReluLikelihood, "h-tanh": T.HardTanhLikelihood, "h-sigm": T.HardSigmoidLikelihood,
           "door": T.SymmetricDoorLikelihood}[c["name"]]
    return cls(y=c["y"], isotropic=False, dtype=dtype, ctx=ctx, **kw)


def close(x, ref, dtype, atol, what):
    x, ref = np.asarray(x, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert x.shape == ref.shape, what
    assert np.array_equal(np.isnan(x), np.isnan(ref)), what
    err = np.abs(x - ref)
    lim = RTOL[dtype] * np.abs(ref) + atol
    bad = ~(err <= lim) & ~np.isnan(ref)
    assert not bad.any(), (what, int(bad.sum()), "worst at", int(np.nanargmax(err - lim)),
                           x.ravel()[np.nanargmax(err - lim)], ref.ravel()[np.nanargmax(err - lim)])


def v_scale(c):
    """the magnitude the component variance is a DIFFERENCE of (so that eps of it is the rounding floor of v, whoever
    evaluates the formula): abs likelihood y^2 (1 - tanh^2), binary prior 1 - tanh^2, truncated normals v0 (1 - ...)"""
    a, n = c["a"], c["name"]
    if c["family"] == "likelihood" and n == "abs":
        return c["y"] ** 2 + 0 * a
    if c["family"] == "prior" and n == "binary":
        return 1.0 + 0 * a
    if c["family"] == "likelihood" and n not in ("gaussian",):
        return 1.0 / a + np.abs(c["v"])
    return np.abs(c["v"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_diagonal_beliefs_match_reference_and_oracle(ctx, golden_diag, dtype):
    import tramp_amd as T
    seen = set()
    k = 400 * EPS[dtype]
    for c in golden_diag:
        if dtype == np.float32 and c["inputs"] == "edge":
            continue                                   # precisions of 1e-12 .. 1e12 next to b of 1e11: outside f32 by construction
        tag = (c["family"], c["name"], c["inputs"], c["kwargs"], np.dtype(dtype).name)
        f = build(T, ctx, c, dtype)
        assert not f.isotropic
        a, b = c["a"], c["b"]
        if c["family"] == "prior":
            r, v = f.compute_forward_posterior(a, b)
            a_new, b_new = f.compute_forward_message(a, b)
        else:
            r, v = f.compute_backward_posterior(a, b, c["y"])
            a_new, b_new = f.compute_backward_message(a, b)
        assert r.shape == b.shape and v.shape == b.shape, tag               # test_priors.py:29-30
        with np.errstate(all="ignore"):
            r_o, v_o = diag_oracle_posterior(orc, c)
        v_o = np.broadcast_to(v_o, b.shape)
        dv = k * v_scale(c)                                                  # rounding floor of v, per component
        for src, ref_r, ref_v in (("reference", c["r"], c["v"]), ("oracle", r_o, v_o)):
            close(r, ref_r, dtype, k * (np.abs(ref_r) + np.sqrt(1.0 / a)), tag + (src, "r"))
            close(v, ref_v, dtype, dv, tag + (src, "v"))
        if c["name"] == "gaussian":                                          # constant messages, whatever comes in
            np.testing.assert_allclose(a_new, c["a_new"], rtol=1e-6 if dtype == np.float32 else 1e-14)
            np.testing.assert_allclose(b_new, c["b_new"], rtol=1e-6 if dtype == np.float32 else 1e-14, atol=1e-300)
        else:
            # a_new = clip(1/v - a): an error dv of v is dv / v^2 of 1/v; b_new = r (a + a_new) - b
            with np.errstate(all="ignore"):
                da = (dv + RTOL[dtype] * np.abs(c["v"]) * (dtype == np.float32)) / np.maximum(c["v"], 1e-20) ** 2
                da = np.nan_to_num(np.minimum(da, 2e11), nan=0.0) + k * (np.abs(a) + np.abs(c["a_new"]))
            close(a_new, c["a_new"], dtype, da, tag + ("a_new",))
            cond_b = np.nan_to_num(np.abs(c["r"]) * (np.abs(a) + np.abs(c["a_new"])) + np.abs(b), nan=0.0)
            close(b_new, c["b_new"], dtype, np.nan_to_num(np.abs(c["r"]), nan=0.0) * da + (k + RTOL[dtype] * (dtype == np.float32)) * cond_b + 1e-300,
                  tag + ("b_new",))
        if getattr(f, "_logz_kind", None) is not None or c["family"] == "prior" and c["name"] != "binary":
            A = f.compute_log_partition(a, b) if c["family"] == "prior" else f.compute_log_partition(a, b, c["y"])
            assert isinstance(A, float)
            np.testing.assert_allclose(A, c["A"], rtol=1e-9 if dtype == np.float64 else 2e-4,
                                       atol=1e-12 if dtype == np.float64 else 1e-5, err_msg=str(tag))
        seen.add((c["family"], c["name"]))
    assert len(seen) == 4 + 8, seen
