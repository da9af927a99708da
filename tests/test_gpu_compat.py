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


RACCOON_STYLE_SCRIPT = """
import numpy as np
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# --- the structure of the reference's examples/sparse/plot_raccoon_denoise.py:82-150 on a small piecewise-constant image ---
from tramp.priors.base_prior import Prior
from tramp.experiments import TeacherStudentScenario
from tramp.priors import GaussianPrior, GaussBernoulliPrior, MAP_L21NormPrior
from tramp.channels import GaussianChannel, GradientChannel
from tramp.variables import SIMOVariable, MILeafVariable, SISOVariable as V, SILeafVariable as O
from tramp.algos import ConstantInit


class BlocksPrior(Prior):                       # a sample-only prior, as the example's RaccoonPrior
    def __init__(self):
        rs = np.random.RandomState(0)
        x = np.kron(rs.standard_normal((4, 4)), np.ones((8, 8)))
        self.x = (x - x.mean()) / x.std()
        self.size = self.x.shape

    def sample(self):
        return self.x


prior = BlocksPrior()
x_shape = prior.size
noise = GaussianChannel(var=0.5)
grad_shape = (2,) + x_shape
teacher = (prior @ SIMOVariable("x", n_next=2) @ (GradientChannel(x_shape) @ O("x'") + noise @ O("y"))).to_model()
sample = teacher.sample()
assert sample["x"].shape == x_shape and sample["x'"].shape == grad_shape and sample["y"].shape == x_shape

sparse_grad = (GaussianPrior(size=x_shape) @ SIMOVariable(id="x", n_next=2) @ (
    noise @ O("y") + (GradientChannel(shape=x_shape) + GaussBernoulliPrior(size=grad_shape, var=0.7, rho=0.9))
    @ MILeafVariable(id="x'", n_prev=2))).to_model()
scenario = TeacherStudentScenario(teacher, sparse_grad, x_ids=["x", "x'"])
scenario.setup(seed=1)
_ = scenario.run_ep(max_iter=100, damping=0.1)
y = scenario.observations["y"]
mse_y = np.mean((y - scenario.x_true["x"]) ** 2)
mse_sg = np.mean((np.asarray(scenario.x_pred["x"]).reshape(x_shape) - scenario.x_true["x"]) ** 2)

tv_denoiser = (GaussianPrior(size=x_shape) @ SIMOVariable(id="x", n_next=2) @ (
    noise @ O("y") + (GradientChannel(shape=x_shape) + MAP_L21NormPrior(size=grad_shape, gamma=1))
    @ MILeafVariable(id="x'", n_prev=2))).to_model()
scenario2 = TeacherStudentScenario(teacher, tv_denoiser, x_ids=["x", "x'"])
scenario2.setup(seed=1)
_ = scenario2.run_ep(max_iter=100, damping=0, initializer=ConstantInit(a=1, b=1))
mse_tv = np.mean((np.asarray(scenario2.x_pred["x"]).reshape(x_shape) - scenario2.x_true["x"]) ** 2)
assert np.shape(scenario.x_pred["x"]) == x_shape and np.shape(scenario.x_pred["x'"]) == grad_shape   # estimates keep the variable's shape
print("mse y / sparse gradient / tv:", mse_y, mse_sg, mse_tv)
assert mse_sg < 0.6 * mse_y and mse_tv < 0.6 * mse_y
print("ok")
"""


def test_the_raccoon_example_structure_runs_on_this_package():
    """examples/sparse/plot_raccoon_denoise.py in small: a sample-only user prior in the teacher, two observed leaves, the sparse-
    gradient and total-variation students through TeacherStudentScenario - written against the reference's import paths"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", RACCOON_STYLE_SCRIPT], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONPATH=root), cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1000:], out.stderr[-3000:])


EXAMPLES_GLM_SCRIPT = """
import numpy as np
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# --- the run_* functions of the reference's examples/glm/data/*.py, bodies as written there, at one or two grid points ---
from tramp.models import glm_generative, glm_state_evolution
from tramp.experiments import run_experiments, BayesOptimalScenario, run_state_evolution, find_critical_alpha
from tramp.algos import EarlyStopping, EarlyStoppingEP, CustomInit


def run_cs(N, alpha, f, prior_rho):                               # cs_universality.py:7-18
    model = glm_generative(
        N=N, alpha=alpha, ensemble_type="random_feature",
        prior_type="gauss_bernoulli", output_type="gaussian",
        ensemble_f=f, prior_rho=prior_rho, output_var=1e-11
    )
    scenario = BayesOptimalScenario(model, x_ids=["x"])
    early = EarlyStopping()
    records = scenario.run_all(
        metrics=["mse"], max_iter=200, callback=early
    )
    return records


def run_perceptron(N, alpha, p_pos):                              # perceptron_ep_vs_se.py:7-16
    model = glm_generative(
        N=N, alpha=alpha,
        ensemble_type="gaussian", prior_type="binary", output_type="sgn",
        prior_p_pos=p_pos
    )
    scenario = BayesOptimalScenario(model, x_ids=["x"])
    early = EarlyStopping()
    records = scenario.run_all(max_iter=200, callback=early)
    return records


def run_phase_retrieval(N, alpha, prior_mean):                    # phase_retrieval_ep_vs_se.py:7-19
    model = glm_generative(
        N=N, alpha=alpha, ensemble_type="complex_gaussian",
        prior_type="gauss_bernoulli", output_type="modulus",
        prior_mean=prior_mean, prior_rho=0.5
    )
    scenario = BayesOptimalScenario(model, x_ids=["x"])
    early = EarlyStopping(wait_increase=10)
    records = scenario.run_all(
        metrics=["mse", "phase_mse"],
        max_iter=200, damping=0.3, callback=early
    )
    return records


def run_critical(a0, prior_rho, criterion):                       # cs_critical_lines.py:6-15 (alpha_tol 0.02 here)
    alpha = find_critical_alpha(
        id="x", a0=a0, mse_criterion=criterion,
        alpha_min=1e-5, alpha_max=2., alpha_tol=0.02,
        model_builder=glm_state_evolution,
        prior_type="gauss_bernoulli", output_type="gaussian",
        prior_rho=prior_rho, output_var=1e-11
    )
    return dict(alpha=alpha)


def run_se_door(a0, alpha, output_width, prior_p_pos):            # door_mse_curves.py:7-21
    model = glm_state_evolution(
        alpha=alpha, prior_type="binary", output_type="door",
        output_width=output_width, prior_p_pos=prior_p_pos
    )
    a_init = [("x", "bwd", a0)]
    initializer = CustomInit(a_init=a_init)
    early = EarlyStopping(max_increase=0.1)
    records = run_state_evolution(
        x_ids=["x"], model=model,
        max_iter = 200, initializer=initializer,
        callback = early
    )
    return records


def run_se_pr(a0, alpha, prior_rho, prior_mean):                  # phase_retrieval_mse_curves.py:7-18
    model = glm_state_evolution(
        alpha=alpha, prior_type="gauss_bernoulli", output_type="modulus",
        prior_rho=prior_rho, prior_mean=prior_mean
    )
    a_init = [("x", "bwd", a0)]
    initializer = CustomInit(a_init=a_init)
    records = run_state_evolution(
        x_ids=["x"], model=model, max_iter=200, initializer=initializer
    )
    return records


def run_se_abs(a0, alpha, prior_rho, prior_mean):                 # sgn_retrieval_mse_curves.py:7-19
    model = glm_state_evolution(
        alpha=alpha, prior_type="gauss_bernoulli", output_type="abs",
        prior_rho=prior_rho, prior_mean=prior_mean
    )
    a_init = [("x", "bwd", a0)]
    initializer = CustomInit(a_init=a_init)
    records = run_state_evolution(
        x_ids=["x", "z"], model=model,
        max_iter = 200, initializer=initializer
    )
    return records


np.random.seed(0)
df = run_experiments(run_cs, N=500, f=["abs", "tanh"], prior_rho=0.25, alpha=[0.8])
assert len(df) == 6 and set(df.source) == {"SE", "EP", "mse"}, df
assert float(df[(df.f == "tanh") & (df.source == "mse")].v.iloc[0]) < 1e-3
# as written the perceptron run fails in the reference too: its state evolution starts the sign likelihood at az = 1 / tau_z
# ("az must be greater than 1/ tau_z", sgn_likelihood.py:81) and run_experiments logs and skips it; the EP half runs
df = run_experiments(run_perceptron, N=500, p_pos=[0.5], alpha=[1.8])
assert len(df) == 0, df
sc = BayesOptimalScenario(glm_generative(N=500, alpha=1.8, ensemble_type="gaussian", prior_type="binary", output_type="sgn",
                                         prior_p_pos=0.5), x_ids=["x"])
rec = {r["source"]: r for r in sc.run_all(source="EP", max_iter=200, callback=EarlyStopping())}
assert rec["mse"]["v"] < 0.5 and rec["EP"]["n_iter"] <= 200
df = run_experiments(run_phase_retrieval, N=200, prior_mean=[1], alpha=[2.5])
assert len(df) == 4 and float(df[df.source == "phase_mse"].v.iloc[0]) < 0.1, df
df = run_experiments(run_critical, a0=0, prior_rho=[0.5], criterion=["perfect"])
assert 0.5 < float(df.alpha.iloc[0]) < 1.0, df
df = run_experiments(run_se_door, a0=0.1, output_width=[1.0], prior_p_pos=0.51, alpha=[1.0, 2.5])
assert len(df) == 2 and np.all(df.v >= 0) and np.all(df.v <= 1.0001), df
df = run_experiments(run_se_pr, a0=[0.1, 1000], prior_rho=[0.6], prior_mean=0, alpha=[2.0])
assert len(df) == 2 and float(df[df.a0 == 1000].v.iloc[0]) < 0.1, df
df = run_experiments(run_se_abs, a0=[1000], prior_rho=[0.4], prior_mean=0, alpha=[1.0])
assert len(df) == 2 and set(df.x_id) == {"x", "z"}, df


def run_critical_door(a0, output_width, prior_p_pos, criterion):  # door_critical_lines.py:7-15 (alpha_tol 0.05 here)
    alpha = find_critical_alpha(
        id="x", a0=a0, mse_criterion=criterion,
        alpha_min=0.1, alpha_max=3., alpha_tol=0.05,
        model_builder=glm_state_evolution,
        prior_type="binary", output_type="door",
        output_width=output_width, prior_p_pos=prior_p_pos
    )
    return dict(alpha=alpha)


df = run_experiments(run_critical_door, a0=[0.1, 1000], output_width=[1.0], prior_p_pos=0.51, criterion=["random", "perfect"])
# from the uninformed start a0 = 0.1: "random" = where the estimate leaves the prior's variance, "perfect" = where it reaches v = 0.
# From a0 = 1000 the binary prior saturates the message at the 1e11 clip, where v = 0 is a stable fixed point at EVERY alpha (the
# information of the door observation grows like sqrt(az) - csrc/se_kernels.hip GainAcc), so neither criterion has a bracket in
# [0.1, 3]: binary_search raises "Bad bounds" and run_experiments skips those, as the reference does whenever its bracket fails
# (its informed runs are decided by the round-off of 1 / v - az at the clip)
assert len(df) == 2 and set(df.a0) == {0.1} and set(df.criterion) == {"random", "perfect"}, df
crit = df.set_index("criterion").alpha
assert 0.1 < crit["random"] < crit["perfect"] < 3.0, df
for a0, alpha, want_perfect in ((0.1, 1.0, False), (0.1, 3.0, True), (1000, 1.0, True), (1000, 3.0, True)):
    v = run_se_door(a0, alpha, 1.0, 0.51)[0]["v"]
    assert (v < 1e-6) == want_perfect, (a0, alpha, v)

# tmp/plot_symmetric_door.py:20-41: the door teacher, its factors printed, EP dynamics on both variables
import matplotlib
matplotlib.use("Agg")
from tramp.experiments import qplot, plot_compare
teacher = glm_generative(
    N=500, alpha=1.6, ensemble_type="gaussian", prior_type="binary",
    output_type="door", output_width=1., prior_p_pos=0.51
)
for factor in teacher.factors:
    print(factor)
scenario = BayesOptimalScenario(teacher, x_ids=["x", "z"])
scenario.setup(seed=42)
scenario.student.plot()
ep_evo = scenario.ep_convergence(
    metrics=["mse", "sign_mse"], damping=0.1, max_iter=40
)
fig = qplot(
    ep_evo, x="iter",
    y=["mse", "sign_mse", "v"],  y_markers=["x", ".", "-"],
    column="id", y_legend=True
)
plot_compare(scenario.x_true["x"], scenario.x_pred["x"])
assert len(fig.get_axes()) == 2 and set(ep_evo.id) == {"x", "z"} and len(ep_evo) == 80
assert np.all(ep_evo.sign_mse <= ep_evo.mse + 1e-12)
print("ok")
"""


def test_the_run_functions_of_the_reference_glm_examples(ctx):
    """examples/glm/data/*.py: their run_* functions with the bodies as written there (random-feature compressed sensing,
    perceptron, phase retrieval with the phase-symmetric mse, the critical line, door / modulus / abs state evolution)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", EXAMPLES_GLM_SCRIPT], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONPATH=root), cwd=root, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1500:], out.stderr[-3000:])


SPARSE_FFT_SCRIPT = """
import numpy as np
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# --- examples/sparse/plot_sparse_fft.py: the teacher class and the student builder as written there (plots left out) ---
from tramp.algos import EarlyStoppingEP
from tramp.variables import SISOVariable as V, SILeafVariable as O, MILeafVariable, SIMOVariable
from tramp.channels import DFTChannel, GaussianChannel
from tramp.priors import GaussBernoulliPrior, GaussianPrior
from tramp.experiments import TeacherStudentScenario


class SparseFFT_Teacher():
    def __init__(self, N, noise_var):
        self.t = np.linspace(-2*np.pi, 2*np.pi, N, endpoint=False)
        self.channel = GaussianChannel(var=noise_var)

    def sample(self, seed=None):
        if seed:
            np.random.seed(seed)
        x = np.cos(self.t) + np.sin(2*self.t)
        z = DFTChannel(real=True).sample(x)
        y = self.channel.sample(x)
        return {"x": x, "z": z, "y": y}


def build_sparse_fft_student(N, prior_var, rho, fft_var, noise_var):
    x_shape = (N,)
    z_shape = (2, N)
    student = (
        GaussianPrior(size=x_shape, var=prior_var) @
        SIMOVariable(id="x", n_next=2) @ (
            GaussianChannel(var=noise_var) @ O("y") + (
                DFTChannel(real=True) +
                GaussBernoulliPrior(size=z_shape, var=fft_var, rho=rho)
            ) @
            MILeafVariable(id="z", n_prev=2)
        )
    ).to_model()
    return student


N, rho, noise_var, seed = 100, 0.02, 0.1, 1
prior_var, fft_var = 1, 18.75
teacher = SparseFFT_Teacher(N, noise_var)
student = build_sparse_fft_student(N, prior_var, rho, fft_var, noise_var)
scenario = TeacherStudentScenario(teacher, student, x_ids=["x", "z"])
scenario.setup(seed=seed)
_ = scenario.run_ep(
    max_iter=1000, damping=0.1, callback=EarlyStoppingEP(tol=1e-2)
)
x, y = scenario.x_true["x"], scenario.observations["y"]
assert scenario.x_pred["x"].shape == (N,) and scenario.x_pred["z"].shape == (2, N) and scenario.x_true["z"].shape == (2, N)
mse_x, mse_y = np.mean((scenario.x_pred["x"] - x) ** 2), np.mean((y - x) ** 2)
print("mse", mse_x, "against", mse_y)
assert mse_x < 0.3 * mse_y
# the two spectral lines of cos(t) + sin(2 t) sampled over two periods: the estimate of z finds them
top = np.argsort(-np.hypot(*scenario.x_pred["z"]))[:4]
assert set(top) == set(np.argsort(-np.hypot(*scenario.x_true["z"]))[:4])
print("ok")
"""


def test_the_sparse_fft_example_runs_on_this_package():
    "examples/sparse/plot_sparse_fft.py: a teacher that is a plain class with sample(), DFTChannel(real=True) in a tree student"
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", SPARSE_FFT_SCRIPT], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONPATH=root), cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1000:], out.stderr[-3000:])


VAE_STYLE_SCRIPT = """
import numpy as np
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# --- the model building and the EP run of examples/vae_prior/plot_vae.py (lines as written there; random decoder weights
#     instead of the trained ones, a prior sample instead of an MNIST digit) ---
from tramp.algos import ExpectationPropagation, TrackEstimate, NoisyInit
from tramp.algos.metrics import mean_squared_error
from tramp.variables import SISOVariable as V, SILeafVariable as O
from tramp.likelihoods import GaussianLikelihood
from tramp.priors import GaussianPrior
from tramp.channels import LinearChannel, ReshapeChannel, LeakyReluChannel, HardTanhChannel, BiasChannel

rs = np.random.RandomState(0)
D, N1, N = 20, 60, 144
W1, W2 = rs.standard_normal((N1, D)) / np.sqrt(D), rs.standard_normal((N, N1)) / np.sqrt(N1)
b1, b2 = 0.1 * rs.standard_normal(N1), 0.1 * rs.standard_normal(N)
shape = (N)


def build_prior():
    prior_x = (GaussianPrior(size=D) @ V(id="z_0") @
               LinearChannel(W1, name="W_1") @ V(id="Wz_1") @ BiasChannel(b1) @ V(id="b_1") @ LeakyReluChannel(0) @ V(id="z_1") @
               LinearChannel(W2, name="W_2") @ V(id="Wz_2") @ BiasChannel(b2) @ V(id="b_2") @ HardTanhChannel() @ V(id="z_2") @
               ReshapeChannel(prev_shape=N, next_shape=shape))
    return prior_x


# sample_from_prior
model = build_prior() @ O(id="y")
model = model.to_model()
x_star = model.sample()["y"]
assert x_star.shape == (N,) and np.abs(x_star).max() <= 1.0

# inpainting: a band of pixels removed
F = np.identity(N)
for rem in range(60, 84):
    F[rem, rem] = 0
F_obs = np.delete(F, np.where(~F.any(axis=0))[0], axis=0)
y = F_obs @ x_star
model = build_prior() @ V(id="x") @ LinearChannel(F_obs, name="F") @ V(id="z")
model = model @ GaussianLikelihood(y=y, var=1e-2)
model = model.to_model_dag()
model = model.to_model()

initializer = NoisyInit()
x_tracker = TrackEstimate(ids="x", every=100)
callback = x_tracker
ep = ExpectationPropagation(model)
ep.iterate(
    max_iter=250,
    callback=callback,
    initializer=initializer,
    damping=0.5
)
df_track = x_tracker.get_dataframe()
ep_x_data = ep.get_variables_data(["x"])
x_pred = {x_id: data["r"] for x_id, data in ep_x_data.items()}
mse_ep = {x_id: data["v"] for x_id, data in ep_x_data.items()}
mse = min(mean_squared_error(x_star, x_pred['x']), mean_squared_error(x_star, -x_pred['x']))
print(f"mse_ep: {float(np.ravel(mse_ep['x'])[0]):.3f} mse: {mse: .3f}")
assert len(df_track) >= 2 and set(df_track.columns) >= {"id", "r", "iter"}
assert mse < 0.5 * np.mean(x_star ** 2)
print("ok")
"""


def test_the_vae_prior_example_structure_runs_on_this_package():
    "examples/vae_prior/plot_vae.py: decoder chain as a prior (Linear, Bias, LeakyRelu(0), HardTanh, Reshape with integer shapes), inpainting"
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", VAE_STYLE_SCRIPT], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONPATH=root), cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1000:], out.stderr[-3000:])


BENCHMARK_SCRIPT = """
import numpy as np
from time import time
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# --- run_benchmark of examples/figures/compute_benchmark.py:17-70 (the script BASELINE.json's metric comes from), its EP and SE
#     branches as written there; the Lasso / pymc3 competitors are not this package's business ---
from tramp.ensembles import GaussianEnsemble
from tramp.priors import GaussBernoulliPrior
from tramp.channels import GaussianChannel, LinearChannel
from tramp.variables import SISOVariable as V, SILeafVariable as O
from tramp.algos.metrics import mean_squared_error
from tramp.experiments import BayesOptimalScenario, run_experiments


def run_benchmark(alpha, algo, seed):
    # create scenario
    N, rho, noise_var = 1000, 0.05, 1e-2
    M = int(alpha*N)
    A = GaussianEnsemble(M=M, N=N).generate()
    t0 = time()
    model = (
        GaussBernoulliPrior(size=N, rho=rho) @ V("x") @
        LinearChannel(A) @ V("z") @ GaussianChannel(var=noise_var) @ O("y")
    ).to_model()
    t1 = time()
    record = {"svd_time": t1 - t0}  # svd precomputation time
    scenario = BayesOptimalScenario(model, x_ids=["x"])
    scenario.setup(seed)
    y = scenario.observations["y"]
    # run algo
    t0 = time()
    if algo == "SE":
        x_data = scenario.run_se(max_iter=1000, damping=0.1)
        record["mse"] = x_data["x"]["v"]
        record["n_iter"] = x_data["n_iter"]
    if algo == "EP":
        x_data = scenario.run_ep(max_iter=1000, damping=0.1)
        x_pred = x_data["x"]["r"]
        record["n_iter"] = x_data["n_iter"]
    t1 = time()
    record["time"] = t1-t0
    if algo != "SE":
        record["mse"] = mean_squared_error(x_pred, scenario.x_true["x"])
    return record


np.random.seed(11)                                  # (the sensing matrices are drawn before setup(seed) reseeds)
df = run_experiments(run_benchmark, algo=["EP", "SE"], alpha=[0.3, 0.6], seed=[1, 2])
assert len(df) == 8 and set(df.columns) == {"svd_time", "mse", "n_iter", "time", "alpha", "algo", "seed"}, df
g = df.assign(mse=df.mse.map(lambda v: float(np.ravel(v)[0]))).groupby(["algo", "alpha"]).mse.mean()
print(g)
assert g["EP", 0.6] < 5e-3 and g["SE", 0.6] < 5e-3 and g["EP", 0.3] > g["EP", 0.6]
for a in (0.3, 0.6):                                                     # EP at N = 1000 follows state evolution
    assert abs(np.log(g["EP", a] / g["SE", a])) < 0.15, (a, g)
assert df[df.algo == "EP"].time.max() < 5.0
print("ok")
"""


def test_the_reference_benchmark_function_runs_as_written():
    "examples/figures/compute_benchmark.py:17-70, EP and SE branches: the function BASELINE.json's metric is defined by"
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", BENCHMARK_SCRIPT], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONPATH=root), cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1500:], out.stderr[-3000:])


SPARSE_EXAMPLES_SCRIPT = """
import numpy as np
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# --- examples/sparse/plot_sparse_gradient.py: teacher class and student builder as written there ---
from tramp.algos import EarlyStoppingEP
from tramp.experiments import TeacherStudentScenario
from tramp.variables import SISOVariable as V, SILeafVariable as O, MILeafVariable, SIMOVariable
from tramp.channels import GaussianChannel, GradientChannel, Blur2DChannel
from tramp.priors import GaussBernoulliPrior, GaussianPrior
from tramp.priors.base_prior import Prior
from tramp.models import MultiLayerModel


class SparseGradTeacher():
    def __init__(self, N, rho, noise_var):
        self.prior_grad = GaussBernoulliPrior(size=(1, N), rho=rho)
        self.channel = GaussianChannel(var=noise_var)

    def sample(self, seed=None):
        if seed:
            np.random.seed(seed)
        x_prime = self.prior_grad.sample()
        x = x_prime.ravel().cumsum()
        x = x - x.mean()
        y = self.channel.sample(x)
        return {"x": x, "z": x_prime, "y": y}


def build_sparse_grad_student(N, rho, noise_var):
    x_shape = (N,)
    z_shape = (1, N)
    student = (
        GaussianPrior(size=x_shape) @
        SIMOVariable(id="x", n_next=2) @ (
            GaussianChannel(var=noise_var) @ O("y") + (
                GradientChannel(shape=x_shape) +
                GaussBernoulliPrior(size=z_shape, rho=rho)
            ) @ MILeafVariable(id="z", n_prev=2)
        )
    ).to_model()
    return student


size, rho, noise_var, seed = 400, 0.04, 1e-2, 1
teacher = SparseGradTeacher(size, rho, noise_var)
student = build_sparse_grad_student(size, rho, noise_var)
scenario = TeacherStudentScenario(teacher, student, x_ids=["x", "z"])
scenario.setup(seed=seed)
_ = scenario.run_ep(
    max_iter=1000, damping=0.1, callback=EarlyStoppingEP(tol=1e-2)
)
x, y = scenario.x_true["x"], scenario.observations["y"]
assert scenario.x_pred["z"].shape == (1, size) and scenario.x_true["z"].shape == (1, size)
assert np.mean((scenario.x_pred["x"] - x) ** 2) < 0.5 * np.mean((y - x) ** 2)

# --- examples/sparse/plot_raccoon_deconv.py:60-73 on a small image: MultiLayerModel with a sample-only prior, Blur2DChannel ---
class BlocksPrior(Prior):
    def __init__(self):
        rs = np.random.RandomState(0)
        x = np.kron(rs.standard_normal((6, 4)), np.ones((8, 12)))
        self.x = (x - x.mean()) / x.std()
        self.size = self.x.shape

    def sample(self):
        return self.x


prior = BlocksPrior()
x_shape = prior.size
blur = Blur2DChannel(shape=x_shape, sigma=[2, 2])
noise = GaussianChannel(var=0.1)
teacher = MultiLayerModel([prior, blur, noise], ids=["x", "z", "y"])
prior = GaussianPrior(size=x_shape)
basic_deconv = MultiLayerModel(
    [prior, blur, noise], ids=["x", "z", "y"]
)
scenario = TeacherStudentScenario(teacher, basic_deconv, x_ids=["x", "z"])
scenario.setup(seed=1)
_ = scenario.run_ep(max_iter=100, damping=0)
assert scenario.x_pred["x"].shape == x_shape and scenario.x_pred["z"].shape == x_shape
mse_z = np.mean((scenario.x_pred["z"] - scenario.x_true["z"]) ** 2)
mse_y = np.mean((scenario.observations["y"] - scenario.x_true["z"]) ** 2)
print("deconvolution: mse of the blurred image", mse_z, "against the observation's", mse_y)
assert mse_z < 0.5 * mse_y
print("ok")
"""


def test_the_sparse_gradient_and_deconvolution_examples_run_on_this_package():
    "examples/sparse/plot_sparse_gradient.py and plot_raccoon_deconv.py (small image): teacher classes, MultiLayerModel, Blur2DChannel"
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", SPARSE_EXAMPLES_SCRIPT], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONPATH=root), cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1500:], out.stderr[-3000:])


PLOT_EXAMPLES_SCRIPT = """
import matplotlib
matplotlib.use("Agg")
import numpy as np
from tramp_amd.compat import install_as_tramp
install_as_tramp()
# --- examples/glm/plot_cs.py:19-39 and plot_complex_pr.py:19-45 as written (N smaller) ---
from tramp.models import glm_generative
from tramp.experiments import BayesOptimalScenario, qplot, plot_compare, plot_compare_complex

np.random.seed(3)                                   # (the examples draw their sensing matrices unseeded)
teacher = glm_generative(
    N=500, alpha=0.8, ensemble_type="gaussian", prior_type="gauss_bernoulli",
    output_type="gaussian", output_var=1e-11, prior_rho=0.5
)
scenario = BayesOptimalScenario(teacher, x_ids=["x"])
scenario.setup(seed=42)
scenario.student.plot()
ep_evo = scenario.ep_convergence(metrics=["mse"], max_iter=10)
fig = qplot(
    ep_evo, x="iter", y=["v", "mse"], y_markers=["-", "."],
    y_legend=True
)
assert 5 <= len(ep_evo) <= 10 and len(fig.get_axes()[0].get_lines()) == 2, (len(ep_evo), len(fig.get_axes()[0].get_lines()), ep_evo)
plot_compare(scenario.x_true["x"], scenario.x_pred["x"])
assert float(ep_evo.mse.iloc[-1]) < 1e-3 * float(ep_evo.mse.iloc[0])

model = glm_generative(
    N=300, alpha=2, ensemble_type="complex_gaussian",
    prior_type="gauss_bernoulli", output_type="modulus",
    prior_mean=0.01, prior_rho=0.5
)
scenario = BayesOptimalScenario(model, x_ids=["x"])
scenario.setup(seed=42)
scenario.student.plot()
ep_evo = scenario.ep_convergence(
    metrics=["mse", "phase_mse"], max_iter=20  # no damping
)
qplot(
    ep_evo, x="iter", y=["phase_mse", "v"],
    y_markers=["x", "-"], y_legend=True
)
ep_evo = scenario.ep_convergence(
    metrics=["mse", "phase_mse"], max_iter=70, damping=0.3
)
qplot(
    ep_evo, x="iter", y=["mse", "phase_mse", "v"],
    y_markers=[".", "x", "-"], y_legend=True
)
plot_compare_complex(scenario.x_true["x"], scenario.x_pred["x"])
assert set(ep_evo.columns) == {"id", "iter", "mse", "phase_mse", "v"} and len(ep_evo) >= 20
print("ok")
"""


def test_the_plot_examples_run_on_this_package():
    "examples/glm/plot_cs.py and plot_complex_pr.py: setup / student.plot() / ep_convergence / qplot / plot_compare as written"
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", PLOT_EXAMPLES_SCRIPT], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONPATH=root), cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1500:], out.stderr[-3000:])
