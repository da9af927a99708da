"""Host-side pieces completed in round 6 (no GPU): every ensemble of the reference's registry (tramp/ensembles/__init__.py:11-21),
the Marchenko-Pastur transforms against values computed by the reference (tests/golden/models2.npz), and the structure of the
remaining total-variation model builders (tramp/models/total_variation_model.py:59-92) - the graphs only, nothing is computed."""
import numpy as np
import pytest


def test_ensemble_registry_matches_the_reference():
    from tramp_amd import ensembles as E
    assert set(E.ENSEMBLE_CLASSES) >= {"gaussian", "complex_gaussian", "rotation", "unitary", "binary", "ternary", "marchenko",
                                       "random_feature", "complex_unitary"}
    np.random.seed(3)
    M, N = 300, 400
    X = E.get_ensemble("binary", M=M, N=N, p_pos=0.3).generate()
    assert X.shape == (M, N) and set(np.unique(X * np.sqrt(N)).round(12)) == {-1.0, 1.0}
    assert abs((X > 0).mean() - 0.3) < 0.01
    X = E.get_ensemble("ternary", M=M, N=N, p_pos=0.2, p_neg=0.5).generate()
    assert set(np.unique(X * np.sqrt(N)).round(12)) == {-1.0, 0.0, 1.0}
    assert abs((X > 0).mean() - 0.2) < 0.01 and abs((X < 0).mean() - 0.5) < 0.01
    R = E.get_ensemble("rotation", N=17).generate()
    np.testing.assert_allclose(R @ R.T, np.eye(17), atol=1e-12)
    assert np.linalg.det(R) == pytest.approx(1.0, abs=1e-10)
    U = E.get_ensemble("unitary", N=9).generate()
    np.testing.assert_allclose(U @ U.conj().T, np.eye(9), atol=1e-12)
    X = E.get_ensemble("complex_unitary", M=7, N=11).generate()
    np.testing.assert_allclose(np.abs(X), 1.0, atol=1e-14)
    X = E.get_ensemble("random_feature", M=200, N=300, f="relu_zero_mean").generate()
    assert X.shape == (200, 300) and abs(X.mean() * np.sqrt(300)) < 0.02          # the activation is centred under N(0, 1)
    with pytest.raises(KeyError):
        E.get_ensemble("random_feature", M=2, N=3, f="nope")
    X = E.get_ensemble("complex_gaussian", M=200, N=300).generate()
    assert np.iscomplexobj(X) and abs(X.real.var() * 300 - 1) < 0.05 and abs(X.imag.var() * 300 - 1) < 0.05


def test_marchenko_pastur_transforms_match_the_reference(golden_models2):
    from tramp_amd.ensembles import MarchenkoPasturEnsemble
    for c in golden_models2["mp"]:
        ens = MarchenkoPasturEnsemble(alpha=c["alpha"])
        assert ens.z_min == pytest.approx(c["z_min"], rel=1e-14) and ens.z_max == pytest.approx(c["z_max"], rel=1e-14)
        assert ens.mean_spectrum == pytest.approx(c["mean_spectrum"], rel=1e-12)
        assert ens.eta_transform(c["gamma"]) == pytest.approx(c["eta"], rel=1e-13)
        assert ens.shannon_transform(c["gamma"]) == pytest.approx(c["shannon"], rel=1e-13, abs=1e-15)
        assert ens.measure(lambda z: z * z) == pytest.approx(c["measure_sq"], rel=1e-10)
        # the Shannon transform is the measure of log(1 + gamma z) (Tulino & Verdu), up to the alpha normalisation the class uses
        if c["alpha"] <= 1:
            direct = ens.measure(lambda z: np.log1p(c["gamma"] * z))
            assert ens.shannon_transform(c["gamma"]) == pytest.approx(direct, rel=1e-6, abs=1e-9)


class _NoDevice:
    "a context stand-in: the model builders only construct factor objects here, nothing is launched"


@pytest.mark.parametrize("builder,kw", [("sparse_gradient_regression", dict(grad_rho=0.1)),
                                        ("sparse_gradient_classification", dict(grad_rho=0.1)),
                                        ("tv_regression", dict(grad_scale=0.5)), ("tv_classification", dict(grad_scale=0.5))])
def test_total_variation_builders_exist_with_the_reference_signatures(builder, kw):
    import inspect
    import tramp_amd as T
    sig = inspect.signature(getattr(T, builder))
    names = list(sig.parameters)
    assert names[:3] == ["A", "y", "x_shape"] and names[3] == list(kw)[0] and names[4:6] == ["noise_var", "prior_var"]


def test_factor_model_rejects_what_the_reference_rejects():
    import tramp_amd as T
    with pytest.raises(TypeError):
        T.FactorModel("not a dag")


def test_duplicate_channel_oracle_matches_the_reference(golden_models2):
    "oracle/tramp_oracle.py's restatement of DuplicateChannel against the vectors computed by the reference"
    from oracle import tramp_oracle as orc
    for c in golden_models2["dup_fn"]:
        r, v = orc.duplicate_posterior(c["az"], c["bz"], c["ax"], c["bx"])
        np.testing.assert_allclose(r, c["rz"], rtol=1e-14)
        assert v == pytest.approx(c["vz"], rel=1e-14) and v == pytest.approx(c["v_bwd_err"], rel=1e-14)
        a, b = orc.duplicate_backward_message(c["az"], c["bz"], c["ax"], c["bx"])
        assert a == pytest.approx(c["a_bwd"], rel=1e-13)
        np.testing.assert_allclose(b, c["b_bwd"], rtol=1e-12, atol=1e-13)
        fa, fb = orc.duplicate_forward_message(c["az"], c["bz"], c["ax"], c["bx"])
        for k in range(c["K"]):
            assert fa[k] == pytest.approx(c["a_fwd"][k], rel=1e-13)
            np.testing.assert_allclose(fb[k], c["b_fwd"][k], rtol=1e-12, atol=1e-13)


def test_metrics_match_the_reference_formulas():
    """tramp/algos/metrics.py:5-41 restated here (mean over all components; sign / phase symmetric minima; overlap) against
    tramp_amd.metrics, one signal and an ensemble of instances"""
    import tramp_amd as T
    rs = np.random.RandomState(0)
    x = rs.randn(2, 50)                                          # a planar complex signal
    p = -x + 0.1 * rs.randn(2, 50)
    assert T.METRICS["mse"](x, p) == pytest.approx(np.mean((x - p) ** 2), rel=1e-15)
    assert T.METRICS["overlap"](x, p) == pytest.approx(np.mean(x * p), rel=1e-15)
    assert T.METRICS["sign_mse"](x, p) == pytest.approx(min(np.mean((x - p) ** 2), np.mean((x + p) ** 2)), rel=1e-15)
    z, zp = x[0] + 1j * x[1], p[0] + 1j * p[1]
    want = min(np.mean(np.abs(z - np.exp(1j * phi) * zp) ** 2) / 2 for phi in np.linspace(0, 2 * np.pi, 100))
    assert T.METRICS["phase_mse"](x, p) == pytest.approx(want, rel=1e-13)
    assert T.METRICS["phase_mse"](x, p) < 0.02 < T.METRICS["mse"](x, p)
    # an ensemble: per-instance scores, the true signal shared or per instance
    P = np.stack([p, x + 0.2 * rs.randn(2, 50), x[::-1].copy()])
    got = T.phase_symmetric_mse(x, P.reshape(3, -1), batch=3)
    assert got.shape == (3,) and got[0] == pytest.approx(want, rel=1e-13)
    np.testing.assert_allclose(T.mean_squared_error(np.stack([x, x, x]), P, batch=3), [np.mean((x - q) ** 2) for q in P], rtol=1e-14)
    np.testing.assert_allclose(T.sign_symmetric_mse(x.ravel(), np.stack([p.ravel(), -p.ravel()])), [T.sign_symmetric_mse(x, p)] * 2, rtol=1e-14)
    with pytest.raises(ValueError):
        T.mean_squared_error(np.zeros(7), np.zeros(10))
    with pytest.raises(KeyError):
        T.TrackErrors({"x": x}, metrics=["nope"])


def test_experiment_grid_and_binary_search():
    "tramp/experiments/multiple_experiments.py and critical_alpha.py:7-29 restated: grids, failures logged and skipped, bisection"
    from tramp_amd import experiments as X
    grid = X.get_experiments_from_kwargs(alpha=[0.5, 1.0], rho=np.array([0.1, 0.2, 0.3]), name="cs")
    assert len(grid) == 6 and grid[0] == dict(alpha=0.5, rho=0.1, name="cs") and grid[-1] == dict(alpha=1.0, rho=0.3, name="cs")
    calls = []

    def run(alpha, rho):
        calls.append((alpha, rho))
        if alpha == 1.0 and rho == 0.2:
            raise RuntimeError("diverged")
        return [dict(source="EP", v=alpha * rho), dict(source="SE", v=alpha + rho)] if rho > 0.1 else dict(source="EP", v=0.0)
    seen = []
    df = X.run_experiments(run, on_progress=lambda i, n: seen.append((i, n)), alpha=[0.5, 1.0], rho=[0.1, 0.2])
    assert len(calls) == 4 and seen[-1] == (4, 4)
    assert len(df) == 1 + 2 + 1 and set(df.columns) == {"source", "v", "alpha", "rho"}          # the failing experiment is skipped
    with pytest.raises(RuntimeError):
        X.simple_run_experiments(run, alpha=[1.0], rho=[0.2])
    res = X.binary_search(lambda x: x > 0.3217, 0.0, 1.0, 1e-6)
    assert res["xmin"] <= 0.3217 <= res["xmax"] and res["xerr"] < 1e-6 and abs(res["xmid"] - 0.3217) < 1e-6
    with pytest.raises(ValueError):
        X.binary_search(lambda x: True, 0.0, 1.0, 1e-3)
    with pytest.raises(ValueError):
        X.TeacherStudentScenario(object(), "not a model")


EXAMPLE_IMPORTS = """
import sys
assert "tramp" not in sys.modules
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# every import line of the reference's examples/ (except the matplotlib helpers qplot / plot_compare of experiments/plots.py)
from tramp.models import glm_generative, glm_state_evolution, MultiLayerModel
from tramp.experiments import save_experiments, run_state_evolution, BayesOptimalScenario, TeacherStudentScenario
from tramp.experiments import run_experiments, find_critical_alpha
from tramp.algos.metrics import mean_squared_error, sign_symmetric_mse
from tramp.algos import EarlyStopping, EarlyStoppingEP, CustomInit, ConstantInit, NoisyInit, StateEvolution
from tramp.algos import ExpectationPropagation, TrackEstimate
from tramp.variables import SISOVariable as V, SILeafVariable as O, MILeafVariable, SIMOVariable
from tramp.priors.base_prior import Prior
from tramp.priors import BinaryPrior, GaussianPrior, GaussBernoulliPrior, MAP_L21NormPrior
from tramp.ensembles import GaussianEnsemble
from tramp.channels import (GaussianChannel, LinearChannel, Blur2DChannel, GradientChannel, DFTChannel, ReshapeChannel,
                            LeakyReluChannel, HardTanhChannel, BiasChannel)
from tramp.likelihoods import GaussianLikelihood
import tramp, tramp_amd
assert tramp is tramp_amd and Prior is tramp_amd.priors.Prior
print("ok")
"""


def test_reference_import_paths_resolve_to_this_package():
    "compat.install_as_tramp(): a script written against the reference's package layout imports this package (own process)"
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", EXAMPLE_IMPORTS], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
    # a different tramp already imported: refused
    code = ("import sys, types; sys.modules['tramp'] = types.ModuleType('tramp')\n"
            "from tramp_amd.compat import install_as_tramp\n"
            "try:\n    install_as_tramp()\nexcept ImportError as e:\n    print('refused')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert out.stdout.strip() == "refused", out.stderr[-2000:]


def test_qplot_and_compare_plots():
    "tramp/experiments/plots.py counterparts: facets, styles, multiple y columns, renaming; Agg backend, nothing shown"
    import matplotlib
    matplotlib.use("Agg")
    import pandas as pd
    from tramp_amd import experiments as X
    rows = [dict(alpha=a, rho=r, source=s, v=a * r + (0.1 if s == "EP" else 0.0), mse=a * r)
            for a in np.linspace(0.1, 1, 5) for r in (0.2, 0.5) for s in ("EP", "SE")]
    df = pd.DataFrame(rows)
    fig = X.qplot(df, x="alpha", y="v", marker="source", column="rho", rename={"alpha": "a", "rho": "r"})
    axs = fig.get_axes()
    assert len(axs) == 2 and [len(ax.get_lines()) for ax in axs] == [2, 2] and axs[0].get_title() == "r=0.2" and axs[0].get_xlabel() == "a"
    assert axs[0].get_lines()[0].get_linestyle() == "None" and axs[0].get_lines()[0].get_marker() == "."
    fig = X.qplot(df[df.source == "SE"], x="alpha", y=["v", "mse"], y_markers=["-", "x"], color="rho", y_legend=True)
    (ax,) = fig.get_axes()
    assert len(ax.get_lines()) == 4 and {l.get_label() for l in ax.get_lines()} == {"rho=0.2 v", "rho=0.2 mse", "rho=0.5 v", "rho=0.5 mse"}
    with pytest.raises(ValueError):
        X.qplot(df, x="alpha", y=["v", "mse"], y_markers=["-"])
    with pytest.raises(ValueError):
        X.qplot(df, x="alpha", y=["v", "mse"], y_markers=["-", "?"])
    rs = np.random.RandomState(0)
    assert len(X.plot_compare(rs.randn(300), rs.randn(300)).get_axes()[0].get_lines()) == 2
    fig = X.plot_compare_complex(rs.randn(2, 50), rs.randn(50) + 1j * rs.randn(50))
    assert [ax.get_xlabel() for ax in fig.get_axes()] == ["phase x true", "modulus x true"]
    X.plot_function(np.tanh)
    matplotlib.pyplot.close("all")
