import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    # The HIP library is built in-tree and git-ignored: a fresh checkout has none.  Build it (hipcc
    # cross-compiles gfx950 without a GPU; a few minutes the first time) rather than fail the C-ABI tests.
    lib = os.path.join(ROOT, "tramp_amd", "libtramp_hip.so")
    if not os.path.exists(lib):
        import subprocess
        res = subprocess.run(["make", "-C", os.path.join(ROOT, "tramp_amd", "csrc")], capture_output=True, text=True)
        if res.returncode != 0:
            raise pytest.UsageError("building tramp_amd/libtramp_hip.so failed:\n" + res.stdout[-2000:] + res.stderr[-4000:])


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    with np.load(path, allow_pickle=True) as f:
        return f["data"].item() if f["data"].shape == () else list(f["data"])


@pytest.fixture(scope="session")
def golden_factors():
    return load_golden("factors")


@pytest.fixture(scope="session")
def golden_linear():
    return load_golden("linear")


@pytest.fixture(scope="session")
def golden_ep():
    return load_golden("ep_traj")


@pytest.fixture(scope="session")
def golden_conv():
    path = os.path.join(GOLDEN, "conv.npz")
    with np.load(path, allow_pickle=True) as f:
        return dict(channels=list(f["data"]), ep=list(f["ep"]))


@pytest.fixture(scope="session")
def golden_pwl3_lik():
    path = os.path.join(GOLDEN, "pwl3_lik.npz")
    with np.load(path, allow_pickle=True) as f:
        return dict(cases=list(f["data"]), se=list(f["se"]), ep=list(f["ep"]))


@pytest.fixture(scope="session")
def golden_f2_logz():
    "log-partitions of the rest of the piecewise-linear family, from the reference (tools/make_golden_f2.py)"
    with np.load(os.path.join(GOLDEN, "f2_logz.npz"), allow_pickle=True) as f:
        return list(f["cases"])


@pytest.fixture(scope="session")
def golden_unitary():
    path = os.path.join(GOLDEN, "unitary.npz")
    with np.load(path, allow_pickle=True) as f:
        return dict(msgs=list(f["data"]), ep=list(f["ep"]))


@pytest.fixture(scope="session")
def golden_logz():
    path = os.path.join(GOLDEN, "logz.npz")
    with np.load(path, allow_pickle=True) as f:
        return dict(cases=list(f["data"]), lin=list(f["lin"]), ep=list(f["ep"]))


@pytest.fixture(scope="session")
def golden_tree():
    path = os.path.join(GOLDEN, "tree.npz")
    with np.load(path, allow_pickle=True) as f:
        return dict(grads=list(f["data"]), unit=list(f["unit"]), ep=list(f["ep"]), reg=list(f["reg"]), gch=list(f["gch"]), se=list(f["se"]), l21=list(f["l21"]), tv=list(f["tv"]), vae=list(f["vae"]), com=list(f["com"]), binp=list(f["binp"]), sgncom=list(f["sgncom"]), comse=list(f["comse"]), binse=list(f["binse"]))


@pytest.fixture(scope="session")
def golden_models2():
    "tools/make_golden_models2.py: the remaining total-variation / classification model builders, Marchenko-Pastur transforms"
    path = os.path.join(GOLDEN, "models2.npz")
    with np.load(path, allow_pickle=True) as f:
        return dict(models=list(f["models"]), mp=list(f["mp"]), dup_fn=list(f["dup_fn"]), dup_ep=list(f["dup_ep"]))


@pytest.fixture(scope="session")
def golden_diag():
    return load_golden("diag")


def diag_oracle_posterior(orc, c):
    """(r, v) arrays of one tests/golden/diag.npz case from the oracle's isotropic=False forms"""
    a, b, kw, n = c["a"], c["b"], c["kwargs"], c["name"]
    if c["family"] == "prior":
        if n == "gaussian":
            return orc.gauss_prior_posterior(a, b, kw.get("mean", 0), kw.get("var", 1))
        if n == "gauss_bernoulli":
            return orc.gb_prior_posterior(a, b, kw.get("rho", 0.5), kw.get("mean", 0), kw.get("var", 1), isotropic=False)
        if n == "binary":
            return orc.binary_prior_posterior(a, b, kw.get("p_pos", 0.5), isotropic=False)
        return orc.l1_prior_posterior(a, b, kw.get("gamma", 1), isotropic=False)
    y = c["y"]
    if n == "gaussian":
        return orc.gauss_lik_posterior(a, b, y, kw.get("var", 1))
    if n == "abs":
        return orc.abs_lik_posterior(a, b, y, isotropic=False)
    if n == "sgn":
        return orc.sgn_lik_posterior(a, b, y, isotropic=False)
    if n in ("relu", "l-relu"):
        return orc.relu_lik_posterior(a, b, y, kw.get("slope", 0.), isotropic=False)
    return orc.pwl3_lik_posterior(n, a, b, y, isotropic=False, param=kw.get("width"))


@pytest.fixture(scope="session")
def golden_se():
    return load_golden("se")


@pytest.fixture(scope="session")
def golden_priors2():
    "Exponential / Positive / GaussianMixture priors from the reference (tools/make_golden_r6.py)"
    with np.load(os.path.join(GOLDEN, "priors2.npz"), allow_pickle=True) as f:
        return dict(cases=list(f["cases"]), se=list(f["se"]), ep=list(f["ep"]))


@pytest.fixture(scope="session")
def golden_complex():
    "ModulusLikelihood, ComplexLinearChannel and phase-retrieval EP from the reference (tools/make_golden_r6.py)"
    with np.load(os.path.join(GOLDEN, "complex.npz"), allow_pickle=True) as f:
        return dict(modulus=list(f["modulus"]), linear=list(f["linear"]), ep=list(f["ep"]), se=list(f["se"]),
                    se_traj=list(f["se_traj"]))


def prior2_oracle(orc, name, kw):
    "(posterior(ax, bx, isotropic=True), factor(size), log_partition(ax, bx)) of one round-6 prior from the oracle"
    if name == "exponential":
        m = kw.get("mean", 1)
        return (lambda ax, bx, isotropic=True: orc.exp_prior_posterior(ax, bx, m, isotropic),
                lambda size: orc.ExpPriorFactor(size, m), lambda ax, bx: orc.exp_prior_log_partition(ax, bx, m))
    if name == "positive":
        return (lambda ax, bx, isotropic=True: orc.positive_prior_posterior(ax, bx, isotropic),
                lambda size: orc.PositivePriorFactor(size), orc.positive_prior_log_partition)
    p, m, v = kw["probs"], kw["means"], kw["vars"]
    return (lambda ax, bx, isotropic=True: orc.gm_prior_posterior(ax, bx, p, m, v, isotropic),
            lambda size: orc.GMPriorFactor(size, p, m, v), lambda ax, bx: orc.gm_prior_log_partition(ax, bx, p, m, v))


def redraw_prior2_W(c):
    "the sensing matrix of a priors2.npz EP case, re-drawn as tools/make_golden_r6.py drew it"
    rs = np.random.RandomState(1000 * c["seed"] + c["N"])
    W = rs.randn(c["M"], c["N"]) / np.sqrt(c["N"])
    assert float(W.sum()) == c["W_sum"]
    return W


def redraw_complex_W(c):
    "the complex sensing matrix of a complex.npz EP case"
    rs = np.random.RandomState(77 * c["seed"] + c["N"])
    M, N = c["M"], c["N"]
    W = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))) / np.sqrt(2 * N)
    assert complex(W.sum()) == c["W_sum"]
    return W


def _gpu_available():
    try:
        from tramp_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    "A device context on GPU 0; GPU tests fail loudly when the HIP library is missing."
    from tramp_amd import device
    return device.default_context()


def regen_W(seed, M, N):
    rs = np.random.RandomState(seed)
    return rs.randn(M, N) / np.sqrt(N)
