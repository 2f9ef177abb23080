"""Plugin surface, part 1: context managers and linear path embeddings.

Host-side mirror of the reference's interface (RudyMorel/shadowing,
shadowing/path_shadowing/path_embedding.py) -- same class names, constructor
arguments, attributes and return conventions, so user code and notebooks keep
working -- written from the behaviour, not from the source:

  ContextManagerBase   4-method protocol                      (ref :13-30)
  PredictionContext    in-context = past, out-context = last `horizon` samples (ref :33-56)
  ImputationContext    a gap of `c` samples between `l` left and `r` right ones (ref :59-88)
  CrossChannelContext  the last channels are out-of-context   (ref :91-114)
  PathEmbedding        linear embedding = conv1d with a (d, 1, K) kernel buffer (ref :117-132)
  Identity(d)          kernel = eye(d): the window itself     (ref :135-139)
  Foveal(alpha, beta, max_context)  multiscale suffix boxes   (ref :142-172)

The MI355X kernels fuse Identity away (a window is read in place from the
trajectory) and take PredictionContext as the integer `horizon`; the classes here
stay ordinary torch modules so that every other combination runs on the generic
torch path of PathShadowing.
"""
from __future__ import annotations

import math
from typing import Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ArrayType = Union[np.ndarray, torch.Tensor]


class ContextManagerBase:
    """Splits a time-series into what is shadowed (in-context) and what is
    predicted (out-context).  Subclasses implement the four methods."""

    def select_in_context(self, x: ArrayType) -> ArrayType:
        raise NotImplementedError

    def select_out_context(self, x: ArrayType) -> ArrayType:
        raise NotImplementedError

    def pad_context(self, x_in_context: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def get_out_times(self):
        raise NotImplementedError


class PredictionContext(ContextManagerBase):
    """Past -> future.  `horizon=None` means there is no out-context at all."""

    def __init__(self, horizon: int | None = None):
        self.horizon = horizon

    def _h(self) -> int:
        return 0 if self.horizon is None else int(self.horizon)

    def select_in_context(self, x: ArrayType) -> ArrayType:
        return x if self.horizon is None else x[..., :-self.horizon]

    def select_out_context(self, x: ArrayType) -> ArrayType:
        return x if self.horizon is None else x[..., -self.horizon:]

    def pad_context(self, x_in_context: torch.Tensor) -> torch.Tensor:
        # zero taps over the future: only windows followed by a full horizon are scanned
        return x_in_context if self.horizon is None else F.pad(x_in_context, (0, self.horizon))

    def get_out_times(self):
        return self._h()


class ImputationContext(ContextManagerBase):
    """`portion = (l, c, r)`: l known samples, a gap of c to impute, r known samples."""

    def __init__(self, portion: Tuple | None = None):
        self.portion = portion

    def select_in_context(self, x: ArrayType) -> ArrayType:
        if self.portion is None:
            return x
        left, _, right = self.portion
        return np.concatenate([x[..., :left], x[..., -right:]], axis=-1)

    def select_out_context(self, x: ArrayType) -> ArrayType:
        if self.portion is None:
            return x
        left, _, right = self.portion
        return x[..., left:-right]

    # the reference spells this method `slect_out_context` (path_embedding.py:70); keep the alias
    slect_out_context = select_out_context

    def pad_context(self, x_in_context: torch.Tensor) -> torch.Tensor:
        if self.portion is None:
            return x_in_context
        left, gap, right = self.portion
        hole = x_in_context.new_zeros(x_in_context.shape[:-1] + (gap,))
        return torch.cat([x_in_context[..., :left], hole, x_in_context[..., -right:]], dim=-1)

    def get_out_times(self):
        return 0 if self.portion is None else self.portion[1]


class CrossChannelContext(ContextManagerBase):
    """The last `out_context_channels` channels are predicted from the others."""

    def __init__(self, out_context_channels: int):
        self.out_context_channels = out_context_channels

    def select_in_context(self, x: ArrayType) -> ArrayType:
        return x[..., : x.shape[-2] - self.out_context_channels, :]

    def select_out_context(self, x: ArrayType) -> ArrayType:
        if self.out_context_channels is None:
            return x
        return x[..., -self.out_context_channels:, :]

    def pad_context(self, x_in_context: torch.Tensor) -> torch.Tensor:
        if self.out_context_channels is None:
            return x_in_context
        shape = list(x_in_context.shape)
        shape[-2] = self.out_context_channels
        return torch.cat([x_in_context, x_in_context.new_zeros(shape)], dim=-2)

    def get_out_times(self):
        return 0


class PathEmbedding(nn.Module):
    """Linear embedding: `forward(x (B,1,T)) -> (B, T-K+1, d)` for a kernel (d,1,K)."""

    def __init__(self, kernel: torch.Tensor):
        super().__init__()
        self.register_buffer("kernel", kernel)

    def adjust_to_context(self, context: ContextManagerBase) -> "PathEmbedding":
        return PathEmbedding(context.pad_context(self.kernel))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # (B, d, T') -> a (B, T', d) VIEW: time stays innermost in memory, which is what
        # fixes the reduction order of the distance downstream ("exactness")
        return F.conv1d(x, self.kernel).transpose(1, 2)


class Identity(PathEmbedding):
    """The window itself (kernel = identity matrix): d = `dimension` = window length."""

    def __init__(self, dimension: int):
        self.d = dimension
        super().__init__(torch.eye(dimension).unsqueeze(1))


class Foveal(PathEmbedding):
    """Multiscale look-back: coordinate i is n_i^(-beta) times the sum of the last
    n_i = int(alpha^(i+1)) samples, for i < dim = floor(ln max_context / ln alpha)."""

    def __init__(self, alpha: float, beta: float, max_context: int, device: str = "cpu"):
        self.alpha, self.beta, self.max_context = alpha, beta, max_context
        self.dim = int(math.floor(np.log(max_context) / np.log(alpha)))
        spans = [int(alpha ** n) for n in range(1, self.dim + 1)]
        self.slices = [slice(-n, None) for n in spans]
        kernel = torch.zeros(self.dim, 1, max_context, dtype=torch.float32, device=device)
        for row, n in enumerate(spans):
            kernel[row, :, -n:] = n ** (-beta)
        super().__init__(kernel)
