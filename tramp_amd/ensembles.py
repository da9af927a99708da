"""Random sensing matrices and analytic spectrum transforms (input generation only).

tramp/ensembles/gaussian_ensemble.py:5-21, tramp/ensembles/marchenko_pastur_ensemble.py:6-53.
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


ENSEMBLE_CLASSES = {"gaussian": GaussianEnsemble, "complex_gaussian": ComplexGaussianEnsemble,
                    "marchenko_pastur": MarchenkoPasturEnsemble}


def get_ensemble(ensemble_type, **kwargs):
    return ENSEMBLE_CLASSES[ensemble_type](**kwargs)
