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


def close(x, ref, dtype, atol):
    np.testing.assert_allclose(np.asarray(x, dtype=np.float64), ref, rtol=RTOL[dtype], atol=atol, equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_diagonal_beliefs_match_reference_and_oracle(ctx, golden_diag, dtype):
    import tramp_amd as T
    seen = set()
    for c in golden_diag:
        if dtype == np.float32 and c["inputs"] == "edge":
            continue                                   # precisions of 1e-12 .. 1e12 next to b of 1e11: outside f32 by construction
        tag = (c["family"], c["name"], c["inputs"], c["kwargs"])
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
        k = 200 * EPS[dtype]
        for ref_r, ref_v in ((c["r"], c["v"]), (r_o, v_o)):
            close(r, ref_r, dtype, k * np.nanmax(np.abs(ref_r)))
            close(v, ref_v, dtype, k * np.nanmax(np.abs(ref_v)) + (1e-6 * np.nanmax(np.abs(ref_v)) if dtype == np.float32 else 0))
        if c["name"] == "gaussian":                                          # constant messages, whatever comes in
            np.testing.assert_allclose(a_new, c["a_new"], rtol=1e-6 if dtype == np.float32 else 1e-14)
            np.testing.assert_allclose(b_new, c["b_new"], rtol=1e-6 if dtype == np.float32 else 1e-14, atol=1e-300)
        else:
            # a_new = clip(1/v - a): conditioned on 1/v = a + a_new; b_new = r (a + a_new) - b on |r| (a + a_new) + |b|
            cond_a = np.abs(a) + np.abs(c["a_new"])
            tol_v = RTOL[dtype] if dtype == np.float32 else 1e-11      # relative error of v feeds 1/v
            close(a_new, c["a_new"], dtype, cond_a * (tol_v + 64 * EPS[dtype]))
            cond_b = np.abs(c["r"]) * cond_a + np.abs(b)
            close(b_new, c["b_new"], dtype, np.nan_to_num(cond_b, nan=0.0) * (tol_v + 1e3 * EPS[dtype]) + 1e-300)
        if getattr(f, "_logz_kind", None) is not None or c["family"] == "prior" and c["name"] != "binary":
            A = f.compute_log_partition(a, b) if c["family"] == "prior" else f.compute_log_partition(a, b, c["y"])
            assert isinstance(A, float)
            np.testing.assert_allclose(A, c["A"], rtol=1e-9 if dtype == np.float64 else 2e-4,
                                       atol=1e-12 if dtype == np.float64 else 1e-5, err_msg=str(tag))
        seen.add((c["family"], c["name"]))
    assert len(seen) == 4 + 8, seen
