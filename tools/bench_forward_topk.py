#!/usr/bin/env python
"""RelativeMSE.forward_topk on a pre-embedded ensemble (N points x d) resident in HBM: the scan kernels
(N one-window paths) against the generic torch formulation on the same device."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import shadowing_amd as sa  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for (S, Tp, d, B, k) in [(2048, 2048, 34, 1, 1024), (2048, 2048, 34, 6, 8192)]:
    y = torch.randn((S, Tp, d), generator=g, device=dev) * 0.02
    x = torch.randn((B, d), generator=g, device=dev) * 0.02
    dist = sa.RelativeMSE()
    out = dist.forward_topk(x, y, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = dist.forward_topk(x, y, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    gen = sa.PathDistance.forward_topk
    ref = gen(dist, x, y, k, n_splits=16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = gen(dist, x, y, k, n_splits=16)
    torch.cuda.synchronize()
    ms_gen = (time.perf_counter() - t0) * 1e3
    same = bool(torch.equal(torch.sort(ref[0], dim=1).values, out[0]))
    print(json.dumps(dict(points=S * Tp, d=d, queries=B, k=k, native_ms=round(ms, 3), generic_torch_ms=round(ms_gen, 3),
                          GBps=round(S * Tp * d * 4 / ms / 1e6, 1), same_distances_as_generic=same)))
