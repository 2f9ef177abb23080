#!/usr/bin/env python
"""Time psh_merge_topk_gathered for G logical shards on one GPU (what every rank runs after the all-gather)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
k, B = 1024, 1
for G in (1, 2, 4, 8):
    rows = 4096
    q = torch.as_tensor(syn.single_query(20, 1)[None, :]).to(dev)
    gathered = torch.empty((G, 3 * B * k), dtype=torch.int32, device=dev)
    for g in range(G):
        ds = torch.as_tensor(syn.dataset(rows, 2048, 10 + g)[:, 0, :]).to(dev)
        out = (gathered[g, :B * k].view(torch.float32).view(B, k), gathered[g, B * k:].view(B, k, 2))
        _native.scan_topk(ds, q, k, h=20, r_offset=g * rows, out=out)
    torch.cuda.synchronize()
    for _ in range(5):
        _native.merge_topk_gathered(gathered, G, B, k, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        _native.merge_topk_gathered(gathered, G, B, k, k)
    e1.record(); torch.cuda.synchronize()
    t_general = e0.elapsed_time(e1) / 50 * 1e3
    a1, i1 = _native.merge_topk_gathered(gathered, G, B, k, k)
    a2, i2 = _native.merge_sorted_gathered(gathered, G, B, k, k)
    assert torch.equal(a1.view(torch.int32), a2.view(torch.int32)) and torch.equal(i1, i2)
    e0.record()
    for _ in range(50):
        _native.merge_sorted_gathered(gathered, G, B, k, k)
    e1.record(); torch.cuda.synchronize()
    print(f"G={G}: {G * k} candidates -> {k}: general merge {t_general:.1f} us, sorted-lists merge {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call")
