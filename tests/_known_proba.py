"""Averaging classes with KNOWN arithmetic, injected in place of un-vendored scatspectra's `Softmax` / `Uniform`
both into the REFERENCE (tests/golden/make_golden.py --predict: its predict() then runs unmodified, PS:256-301) and
into shadowing_amd (the tests that compare with those fixtures).  What the fixtures pin is therefore the reference's
own slicing / axis / call conventions around the classes (PS:245-252), not scatspectra's formulas -- those stay
"parity unpinned" (averaging.py).  Test infrastructure; float64 throughout."""
import numpy as np


class DiscreteProba:
    def __init__(self, weights=None):
        self.weights = weights

    def _w(self, x, axis):
        if self.weights is None:
            return np.full(x.shape, 1.0 / x.shape[axis])
        w = np.asarray(self.weights, dtype=np.float64)
        while w.ndim < x.ndim:
            w = w[..., None]
        return np.broadcast_to(w, x.shape)

    def avg(self, x, axis=0):
        x = np.asarray(x, dtype=np.float64)
        return (self._w(x, axis) * x).sum(axis=axis)

    def std(self, x, axis=0):
        x = np.asarray(x, dtype=np.float64)
        w = self._w(x, axis)
        m = (w * x).sum(axis=axis, keepdims=True)
        return np.sqrt((w * (x - m) ** 2).sum(axis=axis))


class Uniform(DiscreteProba):
    def __init__(self):
        super().__init__(None)


class Softmax(DiscreteProba):
    """weights exp(-d / eta) (NOT a Gaussian: deliberately its own arithmetic), normalised over axis 1."""

    def __init__(self, distances, eta):
        d = np.asarray(distances, dtype=np.float64)
        w = np.ones_like(d) if eta is None else np.exp(-(d - d.min(axis=1, keepdims=True)) / float(eta))
        super().__init__(w / w.sum(axis=1, keepdims=True))
