"""tramp_amd.compat.Backend against the REAL library (the reference itself does not travel to the GPU box): every strict-
mode call the binding of tests/test_compat_reference.py makes - same argument lists - compared with the oracle, i.e. with
what the reference's own compute_*_message methods return (tramp/priors/base_prior.py:13-16, likelihoods/
base_likelihood.py:25-28, channels/base_channel.py:9-17, channels/linear/linear_channel.py:69-117).  f64, 1e-9."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from tramp_amd import compat
    b = compat.Backend()
    yield b
    b.close()


def close(got, ref, scale=None):
    a_g, b_g = got
    a_r, b_r = ref
    assert a_g == pytest.approx(a_r, rel=1e-9)
    sc = np.abs(b_r).max() if scale is None else scale
    np.testing.assert_allclose(b_g, b_r, rtol=1e-9, atol=1e-9 * max(sc, 1e-300))


def test_priors_and_likelihoods(be):
    rs = np.random.RandomState(4)
    n = 1000
    b = rs.standard_normal(n) * 2
    close(be.gb_prior_forward_message(1.3, b, 0.1, 0.2, 1.5), orc.gb_prior_message(1.3, b, rho=0.1, mean=0.2, var=1.5))
    close(be.l1_prior_forward_message(0.8, b, 1.1), orc.l1_prior_message(0.8, b, gamma=1.1))
    close(be.gauss_prior_forward_message(b, 0.3, 2.0), orc.gauss_prior_message(1.0, b, mean=0.3, var=2.0))
    y = rs.standard_normal(n)
    close(be.gauss_lik_backward_message(b, y, 0.05), orc.gauss_lik_message(1.0, b, y, var=0.05))
    ys = np.sign(y)
    close(be.sgn_lik_backward_message(1.7, b, ys), orc.sgn_lik_message(1.7, b, ys))
    close(be.abs_lik_backward_message(1.7, b, np.abs(y)), orc.abs_lik_message(1.7, b, np.abs(y)))
    # shapes survive: a message of shape (4, 5) comes back as (4, 5)
    a_new, b_new = be.gb_prior_forward_message(1.3, b[:20].reshape(4, 5), 0.1, 0.0, 1.0)
    assert b_new.shape == (4, 5) and isinstance(a_new, float)


@pytest.mark.parametrize("kind", ["sgn", "abs", "relu"])
def test_pwl_channels(be, kind):
    from tramp_amd import _lib as L
    rs = np.random.RandomState(6)
    n = 513
    bz, bx = rs.standard_normal(n) * 1.5, rs.standard_normal(n)
    code = dict(sgn=L.SGN, abs=L.ABS, relu=L.RELU)[kind]
    for d, dc in (("fwd", L.FWD), ("bwd", L.BWD)):
        ref = orc.pwl_channel_message(kind, d, 1.3, bz, 0.7, bx)
        close(be.pwl_channel_message(code, dc, 1.3, bz, 0.7, bx), ref, scale=10 * max(np.abs(ref[1]).max(), 1.0))


@pytest.mark.parametrize("shape", [(24, 48), (40, 40), (48, 32)])
def test_linear_channel_from_the_reference_svd(be, shape):
    "factors exactly as the reference's LinearChannel.__init__ holds them (full U, rectangular S, full V: linear_channel.py:8-15)"
    from tramp_amd import _lib as L
    M, N = shape
    rs = np.random.RandomState(M)
    W = rs.randn(M, N) / np.sqrt(N)
    lin = orc.LinearOracle(W)
    h = be.linear_create(lin.U, lin.S, lin.V, lin.rank)
    try:
        bz, bx = rs.standard_normal(N), rs.standard_normal(M)
        for az, ax in ((1.3, 0.7), (2.0, 0.0), (0.4, 5.0)):
            close(be.linear_message(h, L.FWD, az, bz, ax, bx), lin.forward_message(az, bz, ax, bx))
            close(be.linear_message(h, L.BWD, az, bz, ax, bx), lin.backward_message(az, bz, ax, bx))
    finally:
        be.linear_destroy(h)


def test_round6_priors_modulus_and_complex_channel(be):
    """the strict-mode calls the round-6 bindings make (tests/test_compat_reference.py runs the same ones under the
    reference's own iterate()): Binary / Exponential / Positive / GaussianMixture priors, ModulusLikelihood and the
    ComplexLinearChannel built from the reference's complex SVD in its real embedding"""
    from tramp_amd import _lib as L
    rs = np.random.RandomState(8)
    n = 777
    b = rs.standard_normal(n) * 2
    r = np.tanh(b + 0.5 * np.log(0.3 / 0.7))
    close(be.binary_prior_forward_message(1.3, b, 0.3), orc.ab_new(r, (1 - r**2).mean(), 1.3, b))
    close(be.halfline_prior_forward_message(1.3, b, 0.0, -1 / 0.7), orc.exp_prior_message(1.3, b, mean=0.7))
    close(be.halfline_prior_forward_message(0.6, b, 1.0, 0.0), orc.positive_prior_message(0.6, b))
    mix = ([0.2, 0.5, 0.3], [-2.0, 0.0, 1.5], [0.3, 0.01, 2.0])
    close(be.gm_prior_forward_message(1.3, b, *mix), orc.gm_prior_message(1.3, b, *mix))
    bz = rs.standard_normal((2, n)) * 1.5
    y = np.abs(rs.standard_normal(n) + 1j * rs.standard_normal(n))
    got = be.modulus_lik_backward_message(1.7, bz, y)
    assert got[1].shape == (2, n)
    close(got, orc.modulus_lik_message(1.7, bz, y))
    for M, N in ((24, 16), (16, 24)):
        W = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))) / np.sqrt(2 * N)
        lin = orc.ComplexLinearOracle(W)
        h = be.complex_linear_create(lin.U, lin.S, lin.V, lin.rank)
        try:
            cz, cx = rs.standard_normal((2, N)), rs.standard_normal((2, M))
            for az, ax in ((1.3, 0.7), (0.4, 5.0)):
                a, bb = be.linear_message(h, L.FWD, az, cz.reshape(-1), ax, cx.reshape(-1))
                close((a, bb.reshape(2, M)), lin.forward_message(az, cz, ax, cx))
                a, bb = be.linear_message(h, L.BWD, az, cz.reshape(-1), ax, cx.reshape(-1))
                close((a, bb.reshape(2, N)), lin.backward_message(az, cz, ax, cx))
        finally:
            be.linear_destroy(h)


REFERENCE_STYLE_SCRIPT = """
import numpy as np
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# --- from here on the script is written against the reference's API (cf. examples/glm: a sparse-CS sweep over alpha) ---
from tramp.models import glm_generative
from tramp.experiments import BayesOptimalScenario, run_experiments
from tramp.algos import EarlyStopping, EarlyStoppingEP


def run_cs(N, alpha, rho):
    model = glm_generative(N=N, alpha=alpha, ensemble_type="gaussian", prior_type="gauss_bernoulli",
                           output_type="gaussian", prior_rho=rho, output_var=1e-4)
    scenario = BayesOptimalScenario(model, x_ids=["x"])
    return scenario.run_all(metrics=["mse"], max_iter=100, damping=0.1)


np.random.seed(0)
df = run_experiments(run_cs, N=1000, alpha=[0.3, 0.9], rho=0.2)
piv = {(r.alpha, r.source): float(r.v) for r in df.itertuples()}
assert set(df.columns) >= {"source", "x_id", "v", "alpha", "rho", "N"} and len(df) == 6
assert piv[(0.9, "mse")] < 1e-3 < piv[(0.3, "mse")]                      # recovery above the transition, failure below
assert abs(piv[(0.9, "EP")] - piv[(0.9, "SE")]) < 0.5 * piv[(0.9, "SE")] + 1e-5
print("ok")
"""


def test_a_script_written_for_the_reference_runs_on_this_package():
    "compat.install_as_tramp(): an examples/glm-style sweep (BayesOptimalScenario + run_experiments) under the reference's import paths"
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", REFERENCE_STYLE_SCRIPT], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONPATH=root), cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1000:], out.stderr[-3000:])
