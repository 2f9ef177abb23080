"""The three figure helpers the reference's notebooks import next to the scan (`plot_closest`, `plot_shadow`,
`plot_volatility`; reference shadowing/plot_utils.py:8-164, imported by tutorial.ipynb:21-24 and testing.ipynb:134-136).

Plotting is OUT of this build's scope: these exist so that `from shadowing import plot_volatility`
keeps importing and a notebook keeps running.  They draw a plain version of each figure -- the present path, the
+-1 sigma band of the weighted close paths, the predicted volatility cones -- with matplotlib imported on the first call.
The weights are `Softmax(distances, eta)` of whatever averaging.py resolved to (scatspectra's class when installed, the
documented stand-in otherwise).
"""
from __future__ import annotations

import numpy as np

__all__ = ["plot_closest", "plot_shadow", "plot_volatility"]


def _plt():
    try:
        import matplotlib.pyplot as plt
    except Exception as e:  # noqa: BLE001
        raise ImportError("shadowing's plot helpers need matplotlib (imported on call); the scan itself does not") from e
    return plt


def _band(distances, close_paths, eta):
    """mean -+ std over the k close paths (k, 1, T) under the Softmax(distances, eta) weights -> two (T,) arrays."""
    from .averaging import Softmax
    paths = np.asarray(close_paths)[None]                      # (1, k, C, T): the axis-1 convention of predict_from_paths
    proba = Softmax(np.asarray(distances)[None, :, None, None], eta)
    m = np.asarray(proba.avg(paths, axis=1))[0, 0]
    s = np.asarray(proba.std(paths, axis=1))[0, 0]
    return m - s, m + s


def _frame(plt, present, horizon, date, color):
    w = present.shape[-1]
    lim = 1.1 * float(np.abs(present).max())
    ax = plt.gca()
    ax.set_ylim(-lim, lim)
    ax.set_xlim(-w - 2, horizon + 2)
    ax.axhline(0.0, color="black", linewidth=0.5)
    ax.axvline(0.0, color="black", linestyle="--", linewidth=1.5)
    ax.set_xlabel("day")
    ax.legend(loc="lower right", fontsize=8)
    if date is not None:
        ax.set_title(date.strftime("%Y/%m/%d"), color=color)


def plot_closest(dlnx_current, close_paths, num_trajectories: int = 20, color_decay: float = 1.2, date=None, color: str = "blue"):
    plt = _plt()
    present = np.asarray(dlnx_current)
    paths = np.asarray(close_paths)
    w = present.shape[-1]
    horizon = paths.shape[-1] - w
    plt.figure(figsize=(4, 2))
    days = np.arange(-w + 1, horizon + 1)
    for j in range(min(num_trajectories, paths.shape[0]) - 1, -1, -1):          # farthest first, nearest on top
        plt.plot(days, paths[j, 0], color="gray", alpha=float(color_decay) ** (-j), linewidth=0.8)
    plt.plot(days[:w], present, color=color, label="present")
    _frame(plt, present, horizon, date, color)


def plot_shadow(dlnx_current, distances, close_paths, eta, date=None, color="blue"):
    plt = _plt()
    present = np.asarray(dlnx_current)
    w = present.shape[-1]
    horizon = np.asarray(close_paths).shape[-1] - w
    lo, hi = _band(distances, close_paths, eta)
    plt.figure(figsize=(4, 2))
    plt.plot(np.arange(-w + 1, 1), present, color=color, label="present")
    plt.fill_between(np.arange(-w + 1, horizon + 1), lo, hi, color="gray", alpha=0.5, label="shadow")
    _frame(plt, present, horizon, date, color)


def plot_volatility(dlnx_current, vol_predictions, Ts, distances=None, close_paths=None, eta=None, date=None,
                    color="blue", color_vol="black"):
    plt = _plt()
    present = np.asarray(dlnx_current)
    w = present.shape[-1]
    horizon = np.asarray(close_paths).shape[-1] - w if close_paths is not None else int(max(Ts))
    daily = np.asarray(vol_predictions) / np.sqrt(252.0)                          # annualised vol -> daily standard deviation
    plt.figure(figsize=(4, 2))
    plt.plot(np.arange(-w + 1, 1), present, color=color, label="present")
    if distances is not None and close_paths is not None:
        lo, hi = _band(distances, close_paths, eta)
        plt.fill_between(np.arange(-w + 1, 1), lo[:w], hi[:w], color="gray", alpha=0.5, label="shadow")
    for i, T in enumerate(Ts):
        s = np.broadcast_to(daily[i], (int(T) + 1,)) if np.ndim(daily[i]) == 0 else np.asarray(daily[i])[: int(T) + 1]
        plt.fill_between(np.arange(len(s)), -s, s, color=color_vol, alpha=0.1, label="vol prediction" if i == 0 else None)
    _frame(plt, present, horizon, date, color)
