"""Random sensing matrices and analytic spectrum transforms (input generation only).

tramp/ensembles/*.py: every ensemble of the reference's registry (tramp/ensembles/__init__.py:11-21).  Host NumPy: these draw
the inputs of a run, they are not on the iteration path.
"""
import numpy as np
from scipy.integrate import quad


class GaussianEnsemble:
    def __init__(self, M, N):
        self.M = M
        self.N = N

    def generate(self):
        "X ~ iid N(0, 1/N), shape (M, N) (gaussian_ensemble.py:11-21)"
        return np.random.randn(self.M, self.N) / np.sqrt(self.N)


class ComplexGaussianEnsemble:
    "X.real and X.imag ~ iid N(0, 1/N) (tramp/ensembles/complex_gaussian_ensemble.py:6-24)"

    def __init__(self, M, N):
        self.M = M
        self.N = N
        self.GE = GaussianEnsemble(M, N)

    def generate(self):
        return self.GE.generate() + 1j * self.GE.generate()


class MarchenkoPasturEnsemble:
    def __init__(self, alpha):
        self.alpha = alpha
        self.z_max = (1 + np.sqrt(alpha))**2
        self.z_min = (1 - np.sqrt(alpha))**2
        self.mean_spectrum = self.measure(lambda z: z)

    def generate(self, N=1000):
        M = int(self.alpha * N)
        return np.random.randn(M, N) / np.sqrt(N)

    def bulk_density(self, z):
        return np.sqrt((z - self.z_min) * (self.z_max - z)) / (2 * np.pi * z)

    def measure(self, f):
        "marchenko_pastur_ensemble.py:31-37"
        atomic = max(0, 1 - self.alpha) * f(0)
        bulk = quad(lambda z: f(z) * self.bulk_density(z), self.z_min, self.z_max)[0]
        return atomic + bulk

    def compute_F(self, gamma):
        return (np.sqrt(gamma * self.z_max + 1) - np.sqrt(gamma * self.z_min + 1))**2

    def eta_transform(self, gamma):
        "marchenko_pastur_ensemble.py:43-45"
        return 1 - self.compute_F(gamma) / (4 * gamma)


    def shannon_transform(self, gamma):
        "marchenko_pastur_ensemble.py:47-53"
        F = self.compute_F(gamma)
        return np.log(1 + self.alpha * gamma - F / 4) + self.alpha * np.log(1 + gamma - F / 4) - F / (4 * gamma)


class BinaryEnsemble:
    """X = +1/sqrt(N) with probability p_pos, -1/sqrt(N) otherwise, iid, shape (M, N) - what tramp/ensembles/
    binary_ensemble.py:6-29 documents (the reference's constructor reads self.p_pos before setting it and its generate
    scales an integer array in place: both raise)."""

    def __init__(self, M, N, p_pos=0.5):
        self.M, self.N, self.p_pos = M, N, p_pos
        self.p_neg = 1 - p_pos

    def generate(self):
        X = np.random.choice([-1.0, +1.0], size=(self.M, self.N), replace=True, p=[self.p_neg, self.p_pos])
        return X / np.sqrt(self.N)


class TernaryEnsemble:
    """X = +1/sqrt(N), 0, -1/sqrt(N) with probabilities p_pos, 1 - p_pos - p_neg, p_neg (tramp/ensembles/
    ternary_ensemble.py:6-31; the reference's in-place scaling of an integer array raises)."""

    def __init__(self, M, N, p_pos=0.33, p_neg=0.33):
        self.M, self.N, self.p_pos, self.p_neg = M, N, p_pos, p_neg
        self.p_zero = 1 - (p_pos + p_neg)

    def generate(self):
        X = np.random.choice([-1.0, 0.0, +1.0], size=(self.M, self.N), replace=True, p=[self.p_neg, self.p_zero, self.p_pos])
        return X / np.sqrt(self.N)


class RotationEnsemble:
    "R ~ Haar SO(N) (tramp/ensembles/rotation_ensemble.py:5-19)"

    def __init__(self, N):
        self.N = N

    def generate(self):
        from scipy.stats import special_ortho_group
        return special_ortho_group.rvs(self.N)


class UnitaryEnsemble:
    "U ~ Haar U(N) (tramp/ensembles/unitary_ensemble.py:5-19)"

    def __init__(self, N):
        self.N = N

    def generate(self):
        from scipy.stats import unitary_group
        return unitary_group.rvs(self.N)


class ComplexUnitaryEnsemble:
    "X = exp(i phi), phi ~ Unif(0, 2 pi), iid, shape (M, N) (tramp/ensembles/complex_unitary_ensemble.py:5-22; scale is unused there too)"

    def __init__(self, M, N, scale=1):
        self.M, self.N, self.scale = M, N, scale

    def generate(self):
        return np.exp(2 * np.pi * 1j * np.random.uniform(size=(self.M, self.N)))


def _relu(x):
    return np.where(x < 0, 0, x)


RANDOM_FEATURE_ACTIVATIONS = {
    "relu": _relu,
    "relu_zero_mean": lambda x: _relu(x) - 1 / np.sqrt(2 * np.pi),
    "abs_zero_mean": lambda x: np.abs(x) - np.sqrt(2 / np.pi),
    "abs": np.abs, "tanh": np.tanh, "sgn": np.sign}


class RandomFeatureEnsemble:
    "X = f(W Z) / sqrt(N) with Z (N, N) ~ N(0, 1/N), W (M, N) ~ N(0, 1) (tramp/ensembles/random_feature_ensemble.py:28-47)"

    def __init__(self, M, N, f):
        self.M, self.N, self.f = M, N, RANDOM_FEATURE_ACTIVATIONS[f]

    def generate(self):
        Z = np.random.randn(self.N, self.N) / np.sqrt(self.N)
        W = np.random.randn(self.M, self.N)
        return self.f(W @ Z) / np.sqrt(self.N)


# tramp/ensembles/__init__.py:11-21 (its key for the Marchenko-Pastur ensemble is "marchenko"; "marchenko_pastur" kept from round 1)
ENSEMBLE_CLASSES = {"gaussian": GaussianEnsemble, "complex_gaussian": ComplexGaussianEnsemble,
                    "rotation": RotationEnsemble, "unitary": UnitaryEnsemble, "binary": BinaryEnsemble,
                    "ternary": TernaryEnsemble, "marchenko": MarchenkoPasturEnsemble,
                    "marchenko_pastur": MarchenkoPasturEnsemble, "random_feature": RandomFeatureEnsemble,
                    "complex_unitary": ComplexUnitaryEnsemble}


def get_ensemble(ensemble_type, **kwargs):
    return ENSEMBLE_CLASSES[ensemble_type](**kwargs)
