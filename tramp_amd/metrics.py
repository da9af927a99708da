"""Scores of an estimate against the true signal (tramp/algos/metrics.py:5-41), per instance of an ensemble.

Host NumPy on the posterior means the scheduler hands out (TrackErrors); the plain mse of the headline path is also
reduced on the device (ExpectationPropagation.mse).  ``x_true``: [n...] shared by the instances, or [batch, n...].
"""
import numpy as np


def _pair(x_true, x_pred, batch):
    """(true, estimate) as [instances, n].  batch=None: arrays of equal size are ONE signal whatever their shape (the reference's
    meaning); an estimate that is a multiple of the true signal holds that many instances of it.  batch=B: the estimate is
    [B, ...], the true signal [B, ...] or one signal shared by the instances."""
    xp = np.asarray(x_pred, dtype=np.float64)
    xt = np.asarray(x_true, dtype=np.float64)
    if batch is None:
        if xp.size % max(xt.size, 1):
            raise ValueError(f"x_true of shape {xt.shape} does not match the estimate of shape {xp.shape}")
        batch = xp.size // xt.size
    xp = xp.reshape(batch, -1)
    if xt.size == xp.size:
        return xt.reshape(batch, -1), xp
    if xt.size != xp.shape[1]:
        raise ValueError(f"x_true of shape {xt.shape} does not match {batch} instances of {xp.shape[1]} components")
    return np.broadcast_to(xt.reshape(1, -1), xp.shape), xp


def _out(v):
    return float(v[0]) if v.size == 1 else v


def mean_squared_error(x_true, x_pred, batch=None):
    xt, xp = _pair(x_true, x_pred, batch)
    return _out(np.mean((xt - xp) ** 2, axis=1))


def sign_symmetric_mse(x_true, x_pred, batch=None):
    "mean squared error up to a global sign (metrics.py:9-15)"
    xt, xp = _pair(x_true, x_pred, batch)
    return _out(np.minimum(np.mean((xt - xp) ** 2, axis=1), np.mean((xt + xp) ** 2, axis=1)))


def phase_symmetric_mse(x_true, x_pred, batch=None):
    """mean squared error up to a global phase, the minimum over the reference's grid of 100 phases in [0, 2 pi] (metrics.py:18-26).
    Complex signals are planar: (2, n) with the real part first (tramp/utils/misc.py:13-27), flattened or not."""
    xt, xp = _pair(x_true, x_pred, batch)
    if xt.shape[1] % 2:
        raise ValueError("phase_mse needs planar complex signals (2, n)")
    n = xt.shape[1] // 2
    zt = xt[:, :n] + 1j * xt[:, n:]
    zp = xp[:, :n] + 1j * xp[:, n:]
    best = np.full(zp.shape[0], np.inf)
    for phi in np.linspace(0, 2 * np.pi, 100):
        d = zt - np.exp(phi * 1j) * zp
        best = np.minimum(best, np.mean(np.concatenate([d.real, d.imag], axis=1) ** 2, axis=1))
    return _out(best)


def overlap(x_true, x_pred, batch=None):
    xt, xp = _pair(x_true, x_pred, batch)
    return _out(np.mean(xt * xp, axis=1))


METRICS = {"sign_mse": sign_symmetric_mse, "phase_mse": phase_symmetric_mse, "mse": mean_squared_error, "overlap": overlap}
