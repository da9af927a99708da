"""tramp_amd - MI355X-native message-passing hot path of Tree-AMP (sphinxteam/tramp).

The per-iteration factor updates of Expectation Propagation and the State
Evolution recursions run as hand-written HIP kernels for gfx950 behind a C ABI
(include/tramp_hip.h, tramp_amd/libtramp_hip.so); this package is the host-side
mirror of the reference's factor-module API (tramp.priors / tramp.channels /
tramp.likelihoods) plus a minimal scheduler (tramp.algos counterpart).

There is no CPU fallback: without the built HIP library the compute methods raise.
"""
from . import _lib
from .device import Context, DeviceArray, default_context
from .priors import (GaussBernoulliPrior, MAP_L1NormPrior, MAP_L21NormPrior, BinaryPrior, GaussianPrior, ExponentialPrior,
                     PositivePrior, GaussianMixturePrior, get_prior)
from . import switches
from .likelihoods import (GaussianLikelihood, SgnLikelihood, AbsLikelihood, AsymmetricAbsLikelihood, ReluLikelihood,
                          LeakyReluLikelihood, HardTanhLikelihood, HardSigmoidLikelihood,
                          SymmetricDoorLikelihood, ModulusLikelihood, get_likelihood)
from .channels import (LinearChannel, SolveLinearChannel, ComplexLinearChannel, ModulusChannel, SgnChannel, AbsChannel, ReluChannel, GaussianChannel,
                       LeakyReluChannel, AsymmetricAbsChannel, HardTanhChannel, HardSigmoidChannel,
                       SymmetricDoorChannel, ConvChannel, Blur1DChannel, Blur2DChannel,
                       DifferentialChannel, LaplacianChannel, DFTChannel, RotationChannel, GradientChannel, UnitaryChannel, ReshapeChannel, BiasChannel, SumChannel, DuplicateChannel,
                       MarchenkoPasturChannel, AnalyticalLinearChannel, get_channel)
from .variables import SISOVariable, SILeafVariable, SIMOVariable, MISOVariable, MILeafVariable
from .models import (Model, DAG, MultiLayerModel, glm_generative, glm_state_evolution,
                     sparse_gradient_regression, sparse_gradient_classification, tv_regression, tv_classification,
                     FactorModel, committee, soft_committee, sgn_committee)
from .ensembles import (GaussianEnsemble, ComplexGaussianEnsemble, MarchenkoPasturEnsemble, BinaryEnsemble, TernaryEnsemble,
                        RotationEnsemble, UnitaryEnsemble, ComplexUnitaryEnsemble, RandomFeatureEnsemble, get_ensemble)
from .algos import (ExpectationPropagation, StateEvolution, EarlyStoppingEP, EarlyStopping,
                    ConstantInit, CustomInit, NoisyInit, JoinCallback, PassCallback, TrackEvolution, TrackErrors,
                    TrackEstimate, TrackMessages, TrackObjective, TrackOverlaps, LogProgress)

from .metrics import METRICS, mean_squared_error, sign_symmetric_mse, phase_symmetric_mse, overlap

from . import experiments
from .experiments import (TeacherStudentScenario, BayesOptimalScenario, run_state_evolution, run_experiments,
                          simple_run_experiments, save_experiments, binary_search, find_state_evolution_mse, find_critical_alpha)

# the reference's nested module paths (tramp.algos.metrics, tramp.priors.base_prior, ...) under this package's name
import sys as _sys
from . import metrics as _metrics, priors as _priors, likelihoods as _likelihoods, channels as _channels, algos as _algos, models as _models
for _path, _mod in (("algos.metrics", _metrics), ("algos.callbacks", _algos), ("algos.initial_conditions", _algos),
                    ("priors.base_prior", _priors), ("likelihoods.base_likelihood", _likelihoods),
                    ("channels.base_channel", _channels), ("models.base_model", _models)):
    _sys.modules[f"{__name__}.{_path}"] = _mod
    setattr(_sys.modules[f"{__name__}.{_path.rsplit('.', 1)[0]}"], _path.rsplit(".", 1)[1], _mod)

__version__ = "0.1.0"
