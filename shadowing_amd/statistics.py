"""Helper used by the README / tutorial snippets of the reference
(shadowing/statistics.py:5-16): annualised realized variance per maturity.  Takes numpy arrays like the
reference's, and torch tensors on any device (so that `predict(..., cuda=True)` can evaluate it where the
shadowing paths already are)."""
from __future__ import annotations

import numpy as np
import torch


def realized_variance(x, Ts, vol: bool = False):
    """x: (..., T) log-returns; Ts: iterable of maturities (in samples).
    Returns (..., len(Ts)): mean(x^2[..., :T]) * 252 (its square root if vol)."""
    if isinstance(x, torch.Tensor):
        if x.is_cuda and x.dtype == torch.float32:
            # on the HIP device: one launch of the library's reduction (psh_realized_variance) over the rows where they lie
            # -- the out-context VIEW of the gathered paths included, no copy
            from . import _native
            out = _native.realized_variance(x, Ts, vol)
            if out is not None:
                return out
        x2 = x ** 2
        out = torch.stack([x2[..., :int(T)].mean(-1) for T in Ts], dim=-1) * 252
        return out ** 0.5 if vol else out
    x = np.asarray(x)
    out = np.stack([(x[..., :T] ** 2).mean(-1) * 252 for T in Ts], axis=-1)
    return np.sqrt(out) if vol else out


# PathShadowing.predict(cuda=True) may evaluate this statistic on the device tensor of the shadowing paths: both
# branches above compute the same quantity (mean of squares; no ddof / median conventions that differ between the
# two libraries)
realized_variance.accepts_torch = True
