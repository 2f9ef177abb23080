#!/usr/bin/env python
"""Where the time of one PathShadowing.shadow(cuda=True) call goes beyond the kernels
(host -> device query, kernel launches, device -> host results): per-piece wall clock."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from shadowing_amd import _native, synthetic as syn  # noqa: E402
from shadowing_amd.path_embedding import Foveal, Identity, PredictionContext  # noqa: E402
from shadowing_amd.path_distance import RelativeMSE  # noqa: E402
from shadowing_amd.path_shadowing import PathShadowing  # noqa: E402


def clock(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    for name, emb, W, R, k, h, B in (("identity cfg2", Identity(20), 20, 32768, 1024, 20, 1),
                                     ("foveal tutorial", Foveal(1.15, 0.9, 126), 126, 2048, 8192, 252, 6)):
        ds = torch.randn((R, 1, 4096), device=dev) * 0.0126
        x = syn.gbm_log_returns((B, W), 3)
        obj = PathShadowing(emb, RelativeMSE(), ds, PredictionContext(horizon=h))
        print(name, "shadow() total ms:", round(clock(lambda: obj.shadow(x, k=k, cuda=True)), 3))
        xt = torch.tensor(x)[:, None, :]
        print("   _native_scan ms:", round(clock(lambda: obj._native_scan(xt, ds, k)), 3))
        d, idx, _ = obj._native_scan(xt, ds, k)
        print("   gather_paths ms:", round(clock(lambda: _native.gather_paths(ds, idx, W + h)), 3))
        paths = _native.gather_paths(ds, idx, W + h)
        print("   paths.cpu().numpy() ms:", round(clock(lambda: paths.cpu().numpy()), 3), "MB", paths.numel() * 4 / 1e6)
        print("   d/idx .cpu() ms:", round(clock(lambda: (d.cpu().numpy(), idx.cpu().numpy())), 3))
        print("   upload q ms:", round(clock(lambda: xt[:, 0, :].contiguous().to(dev)), 3))
        print("   embed q (host conv1d) ms:", round(clock(lambda: emb(xt)), 3))


if __name__ == "__main__":
    main()
