"""Averaging operators of predict_from_paths().

In the reference these come from the un-vendored dependency `scatspectra`
(scattering_spectra v2.0.2: Softmax, Uniform, DiscreteProba; call sites
path_shadowing.py:227-230, :251-252).  When that package is importable it is used
as is.  Otherwise the classes below stand in.  PARITY UNPINNED: the reference
holds no test or golden value at this boundary, and the dependency's source is
not available here; the only in-repo hint is plot_utils.py:65 ("eta: the width of
a Gaussian in the Gaussian average"), i.e. weights proportional to
exp(-d^2 / (2 eta^2)).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - not installable in the build image
    from scatspectra import DiscreteProba, Softmax, Uniform  # type: ignore
    HAVE_SCATSPECTRA = True
except Exception:  # noqa: BLE001
    HAVE_SCATSPECTRA = False

    class DiscreteProba:
        """Weights over an axis of samples; `avg` / `std` are the weighted moments."""

        def __init__(self, weights: np.ndarray | None = None):
            self.weights = weights

        def _w(self, x: np.ndarray, axis: int) -> np.ndarray:
            if self.weights is None:
                return np.full_like(x, 1.0 / x.shape[axis], dtype=np.float64)
            w = np.asarray(self.weights, dtype=np.float64)
            while w.ndim < x.ndim:
                w = w[..., None]
            return np.broadcast_to(w, x.shape)

        def avg(self, x: np.ndarray, axis: int = 0) -> np.ndarray:
            x = np.asarray(x, dtype=np.float64)
            return (self._w(x, axis) * x).sum(axis=axis)

        def std(self, x: np.ndarray, axis: int = 0) -> np.ndarray:
            x = np.asarray(x, dtype=np.float64)
            w = self._w(x, axis)
            m = (w * x).sum(axis=axis, keepdims=True)
            return np.sqrt((w * (x - m) ** 2).sum(axis=axis))

    class Uniform(DiscreteProba):
        def __init__(self):
            super().__init__(None)

    class Softmax(DiscreteProba):
        """Gaussian weights exp(-d^2 / (2 eta^2)), normalised over axis 1 (the k paths)."""

        def __init__(self, distances: np.ndarray, eta: float | None):
            d = np.asarray(distances, dtype=np.float64)
            if eta is None:
                w = np.ones_like(d)
            else:
                z = -(d ** 2) / (2.0 * float(eta) ** 2)
                z = z - z.max(axis=1, keepdims=True)
                w = np.exp(z)
            super().__init__(w / w.sum(axis=1, keepdims=True))
