"""Tuning build: the batched long-window scan with parts switched off, several PSH_DBG masks in ONE process.
usage: lq_ablate2.py W B dbg[,dbg...]   (1 no MFMA chains, 2 no tests, 4 no survivor handling, 8 survivors queued but not verified,
32 the min tree alone; results invalid except for 0 and 16)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from shadowing_amd import _native as N, synthetic as syn
dev = torch.device("cuda:0")
ds = torch.as_tensor(syn.dataset(32768, 4096, 2024)[:, 0, :].copy()).to(dev)
ws = N.Workspace(dev)
for spec in sys.argv[1:]:
    W, B, dbgs = spec.split(":")
    W = int(W); B = int(B)
    q = torch.as_tensor(syn.rolling_queries(B, W, 2025)).to(dev)
    for dbg in dbgs.split(","):
        os.environ["PSH_DBG"] = dbg
        N.scan_topk(ds, q, 1024, h=0, workspace=ws); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            _, _, st, prof = N.scan_topk(ds, q, 1024, h=0, workspace=ws, profile=True)
            best = min(best, prof["scan_ms"])
        print("W=%d B=%d PSH_DBG=%s sample %.3f scan %.3f ms (best of 3) candidates %d" % (W, B, dbg, prof["sample_ms"], best, prof["n_candidates"]), flush=True)
