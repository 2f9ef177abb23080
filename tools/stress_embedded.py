#!/usr/bin/env python
"""Randomised A/B of the embedded scan on the GPU: the suffix-rows fast path against the dense chains
(PSH_FLAG_EMBED_DENSE), bit for bit, over random Foveal-like kernels (with and without an imputation gap), ensemble
shapes (ragged segments, unaligned rows), k, horizons and batch sizes (both block-size instantiations)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from shadowing_amd import _native, synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(n_cases):
    K = int(rng.integers(1, 257))
    alpha = float(rng.uniform(1.05, 2.5))
    dim = max(1, int(np.floor(np.log(max(K, 2)) / np.log(alpha))))
    dim = min(dim, 128, 8192 // ((K + 3) & ~3))
    ker = np.zeros((dim, K), np.float32)
    for i in range(dim):
        n = min(K, max(1, int(alpha ** (i + 1))))
        ker[i, K - n:] = np.float32(n ** (-float(rng.uniform(0.3, 1.2)))) * (1 if rng.random() < 0.8 else -1)
    if rng.random() < 0.4 and K > 8:                               # an ImputationContext's gap
        g0 = int(rng.integers(1, K - 3)); g1 = min(K - 1, g0 + int(rng.integers(1, 9)))
        ker[:, g0:g1] = 0
    if rng.random() < 0.3:
        ker[rng.integers(0, dim)] = 0
    T = int(rng.integers(K + 40, 3000))
    h = int(rng.integers(0, min(30, T - K - 5)))
    R = int(rng.choice([600, 1500, 4096]))
    B = int(rng.choice([1, 2, 5, 7, 13]))
    Tp = T - K - h + 1
    k = int(min(rng.choice([1, 17, 300, 2000]), R * Tp // 4))
    ds = syn.dataset(R, T, 1000 + case)
    # (no NaN in the ensemble here: since round 5 PSH_FLAG_EMBED_DENSE multiplies EVERY tap -- a zero tap over a NaN is a NaN, as in
    #  the reference's conv -- while the fast path visits a row's non-zero span; dirty ensembles behind a linear embedding go
    #  through the clean / dirty split of PathShadowing: tests/test_gpu_nonfinite.py)
    rng.random(); rng.integers(0, R); rng.integers(0, T)          # (the draws of the former NaN case: the other cases stay as they were)
    x = syn.gbm_log_returns((B, K), 2000 + case)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].contiguous()
    dsd, kd, hd = torch.tensor(ds[:, 0, :]).to(dev), torch.tensor(ker).to(dev), hx.to(dev)
    outs = []
    for mode in ("fast", "dense"):
        fl = _native.FLAG_EMBED_DENSE if mode == "dense" else 0
        d, idx, st = _native.scan_topk_embedded(dsd, kd, hd, k, h=h, flags=fl)
        if int(st.max()) != 0:
            d, idx, _ = _native.scan_topk_embedded(dsd, kd, hd, k, h=h, exhaustive=True)
        torch.cuda.synchronize()
        outs.append((d.cpu().numpy(), idx.cpu().numpy(), int(st.max())))
    same = np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32)) and np.array_equal(outs[0][1], outs[1][1])
    print(f"case {case:3d} K={K:3d} d={dim:3d} T={T:4d} h={h:2d} R={R} B={B:2d} k={k:4d} status fast/dense {outs[0][2]}/{outs[1][2]}  {'ok' if same else 'MISMATCH'}")
    bad += 0 if same else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
