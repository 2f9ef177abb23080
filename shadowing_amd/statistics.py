"""Host helper used by the README / tutorial snippets of the reference
(shadowing/statistics.py:5-16): annualised realized variance per maturity."""
from __future__ import annotations

import numpy as np


def realized_variance(x: np.ndarray, Ts, vol: bool = False) -> np.ndarray:
    """x: (..., T) log-returns; Ts: iterable of maturities (in samples).
    Returns (..., len(Ts)): mean(x^2[..., :T]) * 252 (its square root if vol)."""
    x = np.asarray(x)
    out = np.stack([(x[..., :T] ** 2).mean(-1) * 252 for T in Ts], axis=-1)
    return np.sqrt(out) if vol else out
