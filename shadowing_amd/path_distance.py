"""Plugin surface, part 2: distances between embedded paths.

Mirror of the reference's shadowing/path_shadowing/path_distance.py:
  PathDistance.forward(x, y)            broadcast contract: x (..., d), y (..., d) -> (...)   (ref :51-59)
  PathDistance.forward_topk(x, y, k, n_splits)   k smallest over a pre-embedded y           (ref :10-49)
  RelativeMSE                            ||x - y|| / ||x||  (a relative L2 norm)              (ref :62-65)
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn


class PathDistance(nn.Module):

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """x and y broadcast against each other over every dim but the last (the
        embedding dim), which is reduced."""
        raise NotImplementedError

    def forward_topk(self, x: torch.Tensor, y: torch.Tensor, k: int, n_splits: int = 1
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
        """k smallest distances between each x (B1, d) and every y[b2, ...] of a
        pre-embedded y (B2, ..., d), scanning y in `n_splits` chunks of its first dim.

        Returns (B1, k) distances ascending and (B1, k, y.ndim-1) int64 indices into
        y's leading dims.  (The reference builds the index table on the host with
        itertools.product, path_distance.py:36-37; here it is decoded from the flat
        top-k position.)
        """
        n1 = x.shape[0]
        lead = tuple(y.shape[:-1])
        inner = lead[1:]
        n_inner = 1
        for s in inner:
            n_inner *= s
        best_d = x.new_full((n1, k), float("inf"))
        best_i = torch.full((n1, k, len(lead)), 2147483647, dtype=torch.int64, device=x.device)
        xq = x.reshape((n1,) + (1,) * len(lead) + (x.shape[-1],))
        chunk = y.shape[0] // n_splits
        for rows in torch.arange(y.shape[0], device=x.device).split(chunk):
            dist = self(xq, y[rows].unsqueeze(0)).reshape(n1, -1)
            flat = torch.arange(dist.shape[1], device=x.device)
            coords = [rows[flat // n_inner]]
            rem = flat % n_inner
            stride = n_inner
            for s in inner:
                stride //= s
                coords.append(rem // stride)
                rem = rem % stride
            table = torch.stack(coords, dim=-1).expand(n1, -1, -1)
            pool_d = torch.cat([best_d, dist], dim=1)
            pool_i = torch.cat([best_i, table], dim=1)
            best_d, pos = torch.topk(pool_d, k=k, dim=-1, largest=False)
            best_i = torch.gather(pool_i, 1, pos.unsqueeze(-1).expand(-1, -1, len(lead)))
        return best_d, best_i


class RelativeMSE(PathDistance):
    """||x - y||_2 / ||x||_2 over the last dim."""

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return torch.linalg.vector_norm(x - y, dim=-1) / torch.linalg.vector_norm(x, dim=-1)
