"""Teacher-student scenarios, experiment grids and the critical-alpha search of tramp/experiments (teacher_student_scenario.py,
multiple_experiments.py, critical_alpha.py) on top of this package's scheduler.  Host orchestration only - the work is the EP /
SE runs underneath.  One thing the device changes: ``setup(instances=B)`` draws B teacher samples and hands them to ONE
student as an ensemble, so B experiments on the same sensing matrix are one EP run (the reference loops over seeds).
"""
import itertools
import logging

import numpy as np

from .algos import (ExpectationPropagation, StateEvolution, TrackErrors, TrackEvolution, JoinCallback, CustomInit)
from .metrics import METRICS
from .models import Model

from .plots import qplot, plot_compare, plot_compare_complex, plot_function  # noqa: F401  (tramp.experiments exports them)

logger = logging.getLogger(__name__)


class TeacherStudentScenario:
    """tramp/experiments/teacher_student_scenario.py:10-141: ``teacher`` (anything with ``.sample(seed)``) generates the
    signals ``x_ids`` and the observations ``y_ids``; ``student`` (a generative Model with the same ids) infers them."""

    def __init__(self, teacher, student, x_ids=("x",), y_ids=("y",)):
        if not isinstance(student, Model):
            raise ValueError("student not a Model")
        try:
            sample = teacher.sample()
        except AttributeError:
            raise ValueError("teacher does not have a .sample() method")
        for kind, ids in (("x_id", x_ids), ("y_id", y_ids)):
            for vid in ids:
                if vid not in student.variable_ids:
                    raise ValueError(f"{kind} = {vid} not in student variable_ids")
                if vid not in sample:
                    raise ValueError(f"{kind} = {vid} not in teacher variable_ids")
        self.x_ids, self.y_ids = list(x_ids), list(y_ids)
        self.teacher, self.generative_student = teacher, student

    def setup(self, seed=0, instances=1):
        """the teacher generates the data, the student observes it (:45-53).  instances > 1: that many samples (seeds seed + 1,
        seed + 2, ... - seed 0 means "do not reseed", as the reference's Model.sample), stacked on the ensemble axis."""
        if instances == 1:
            sample = self.teacher.sample(seed)
            self.true_values = sample
        else:
            draws = [self.teacher.sample(seed + 1 + k) for k in range(instances)]
            self.true_values = {k: np.stack([np.asarray(d[k]) for d in draws]) for k in draws[0]}
        self.instances = instances
        self.x_true = {vid: self.true_values[vid] for vid in self.x_ids}
        self.observations = {vid: self.true_values[vid] for vid in self.y_ids}
        self.student = self.generative_student.to_observed(self.observations)

    def run_se(self, **algo_kwargs):
        se = StateEvolution(self.student)
        se.iterate(**algo_kwargs)
        x_data = se.get_variables_data(self.x_ids)
        x_data["n_iter"] = se.n_iter
        return x_data

    def run_ep(self, **algo_kwargs):
        ep = ExpectationPropagation(self.student)
        ep.iterate(**algo_kwargs)
        self.ep = ep
        x_data = ep.get_variables_data(self.x_ids)
        for vid in self.x_ids:
            # the estimates in the variable's own shape, as the reference hands them out (an image stays an image, a planar
            # complex vector (2, n)); an ensemble of instances keeps its leading axis
            shape = tuple(ep.shapes[vid])
            r = np.asarray(x_data[vid]["r"])
            x_data[vid]["r"] = r.reshape(shape) if r.size == int(np.prod(shape)) else r.reshape((-1,) + shape)
        x_data["n_iter"] = ep.n_iter
        self.x_pred = {vid: x_data[vid]["r"] for vid in self.x_ids}
        return x_data

    def compute_score(self, x_pred, metrics=("mse",)):
        batch = self.instances if getattr(self, "instances", 1) > 1 else None
        return {vid: {m: METRICS[m](self.x_true[vid], x_pred[vid], batch=batch) for m in metrics} for vid in self.x_ids}

    def run_all(self, source="EP,SE", metrics=("mse",), seed=0, instances=1, **algo_kwargs):
        "the variances predicted by SE and reached by EP, and EP's scores against the truth, as records (:55-82)"
        self.setup(seed=seed, instances=instances)
        records = []
        if "SE" in source:
            x_data = self.run_se(**algo_kwargs)
            records += [dict(source="SE", x_id=vid, v=_scalar(x_data[vid]["v"]), n_iter=x_data["n_iter"]) for vid in self.x_ids]
        if "EP" in source:
            x_data = self.run_ep(**algo_kwargs)
            records += [dict(source="EP", x_id=vid, v=_scalar(x_data[vid]["v"]), n_iter=x_data["n_iter"]) for vid in self.x_ids]
            score = self.compute_score({vid: x_data[vid]["r"] for vid in self.x_ids}, metrics=metrics)
            records += [dict(source=m, x_id=vid, v=_scalar(score[vid][m])) for m in metrics for vid in self.x_ids]
        return records

    def ep_convergence(self, metrics=("mse",), **algo_kwargs):
        "per-iteration variance and scores of an EP run as one DataFrame (:98-117)"
        import pandas as pd
        track, evo = TrackErrors(true_values=self.x_true, metrics=metrics), TrackEvolution(ids=self.x_ids)
        callbacks = [track, evo] + ([algo_kwargs["callback"]] if "callback" in algo_kwargs else [])
        algo_kwargs["callback"] = JoinCallback(callbacks)
        try:
            self.run_ep(**algo_kwargs)
        except Exception as e:      # noqa: BLE001  (the reference logs and returns what was tracked)
            logger.error(e)
        df = pd.merge(track.get_dataframe(), evo.get_dataframe(), on=["id", "iter"])
        for col in ["v"] + list(metrics):
            df[col] = df[col].map(lambda t: np.clip(t, 0, 2))
        return df

    def se_convergence(self, **algo_kwargs):
        "per-iteration variance of an SE run (:119-133)"
        evo = TrackEvolution(ids=self.x_ids)
        callbacks = [evo] + ([algo_kwargs["callback"]] if "callback" in algo_kwargs else [])
        algo_kwargs["callback"] = JoinCallback(callbacks)
        try:
            self.run_se(**algo_kwargs)
        except Exception as e:      # noqa: BLE001
            logger.error(e)
        df = evo.get_dataframe()
        df["v"] = df["v"].map(lambda t: np.clip(t, 0, 2))
        return df


class BayesOptimalScenario(TeacherStudentScenario):
    "the same generative model for teacher and student (:144-156)"

    def __init__(self, model, x_ids=("x",), y_ids=("y",)):
        super().__init__(teacher=model, student=model, x_ids=x_ids, y_ids=y_ids)


def _scalar(v):
    v = np.asarray(v)
    return float(v.reshape(-1)[0]) if v.size == 1 else v


def run_state_evolution(x_ids, model, **algo_kwargs):
    "records (x_id, v, n_iter) of a state-evolution run (:159-178)"
    se = StateEvolution(model)
    se.iterate(**algo_kwargs)
    x_data = se.get_variables_data(ids=x_ids)
    return [dict(x_id=vid, v=_scalar(x_data[vid]["v"]), n_iter=se.n_iter) for vid in x_ids]


# ---------------------------------------------------------------- grids of experiments (multiple_experiments.py)
def log_on_progress(i, total):
    logger.info(f"experiment {i}/{total}")


def _as_list(x):
    return x if isinstance(x, list) else list(x) if isinstance(x, np.ndarray) else [x]


def get_experiments_from_kwargs(**kwargs):
    "the Cartesian product of the (listified) keyword values, one dict per experiment"
    keys = list(kwargs)
    return [dict(zip(keys, values)) for values in itertools.product(*[_as_list(kwargs[k]) for k in keys])]


def _run_grid(run, on_progress, swallow, kwargs):
    import pandas as pd
    experiments = get_experiments_from_kwargs(**kwargs)
    records = []
    for idx, experiment in enumerate(experiments):
        try:
            results = run(**experiment)
            results = [results] if isinstance(results, dict) else results
            for result in results:
                result.update(experiment)
            records += results
        except Exception as e:      # noqa: BLE001
            if not swallow:
                raise
            logger.error(f"Experiment {experiment} failed\n{e}")
        if on_progress:
            on_progress(idx + 1, len(experiments))
    return pd.DataFrame(records)


def run_experiments(run, on_progress=None, **kwargs):
    "run(**experiment) over the grid; failures are logged and skipped (:30-48)"
    return _run_grid(run, on_progress or log_on_progress, True, kwargs)


def simple_run_experiments(run, **kwargs):
    "the same, raising on the first failure (:51-66)"
    return _run_grid(run, None, False, kwargs)


def save_experiments(run, csv_file, on_progress=None, **kwargs):
    run_experiments(run, on_progress, **kwargs).to_csv(csv_file, index=False)


# ---------------------------------------------------------------- critical alpha (critical_alpha.py)
def binary_search(f, xmin, xmax, xtol):
    "bisection on a boolean f with f(xmin) = 0, f(xmax) = 1 (:7-29)"
    ymin, ymax = f(xmin), f(xmax)
    if not (ymin == 0 and ymax == 1):
        raise ValueError(f"Bad bounds: ymin={ymin} and ymax={ymax}")
    max_iter = int(np.log2((xmax - xmin) / xtol)) + 2
    for n_iter in range(1, max_iter + 1):
        xmid = (xmin + xmax) / 2
        ymid = f(xmid)
        xerr = xmax - xmin
        logger.info(f"binary search {n_iter}/{max_iter} xerr={xerr}")
        if xerr < xtol:
            break
        if ymid == 0:
            xmin = xmid
        else:
            xmax = xmid
    assert xerr < xtol
    return dict(xmid=xmid, xmin=xmin, xmax=xmax, xerr=xerr, n_iter=n_iter)


def find_state_evolution_mse(id, a0, alpha, model_builder, **model_kwargs):
    "the variance state evolution reaches for the model at measurement density alpha, from the initial message a0 (:32-57)"
    model = model_builder(alpha=alpha, **model_kwargs)
    se = StateEvolution(model)
    se.iterate(max_iter=200, initializer=CustomInit(a_init=[(id, "bwd", a0)]))
    return _scalar(se.get_variable_data(id=id)["v"])


def find_critical_alpha(id, a0, mse_criterion, alpha_min, alpha_max, model_builder, alpha_tol=1e-6, vtol=1e-3, **model_kwargs):
    """the smallest alpha at which the criterion on the SE variance holds (:60-109): "perfect" (v = 0), "random" (v leaves the
    prior's second moment) or a function v -> bool that is False below alpha_c and True above"""
    if mse_criterion == "perfect":
        def criterion(v):
            return abs(v) < vtol
    elif mse_criterion == "random":
        model = model_builder(alpha=0.5, **model_kwargs)
        model.init_second_moments()
        tau_x = model.get_second_moments()[id]

        def criterion(v):
            return abs(v - tau_x) > vtol
    else:
        criterion = mse_criterion
    return binary_search(lambda alpha: criterion(find_state_evolution_mse(id, a0, alpha, model_builder, **model_kwargs)),
                         alpha_min, alpha_max, alpha_tol)["xmid"]
