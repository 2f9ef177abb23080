"""Tuning build: the batched long-window scan with parts switched off (PSH_DBG: 1 no MFMA chains, 2 no tests; results invalid)."""
import os, sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from shadowing_amd import _native as N, synthetic as syn
dev = torch.device("cuda:0")
W = int(sys.argv[1]); B = int(sys.argv[2])
ds = torch.as_tensor(syn.dataset(32768, 4096, 2024)[:, 0, :].copy()).to(dev)
ws = N.Workspace(dev)
q = torch.as_tensor(syn.rolling_queries(B, W, 2025)).to(dev)
N.scan_topk(ds, q, 1024, h=0, workspace=ws); torch.cuda.synchronize()
_, _, st, prof = N.scan_topk(ds, q, 1024, h=0, workspace=ws, profile=True)
print("PSH_DBG=%s W=%d B=%d sample %.3f scan %.3f ms" % (os.environ.get("PSH_DBG", "0"), W, B, prof["sample_ms"], prof["scan_ms"]))
if int(os.environ.get("PSH_DBG", "0")) & 16:
    lay = N.candidates_layout(32768, 4096, B, W, 0, 1024, ws.buf.numel())
    bc = ws.buf[lay["bcount"]:lay["bcount"] + 4 * lay["max_blocks"]].view(torch.int32)
    print("survivors verified in the last call (all queries): %d = %.1f per (segment, query); admitted %d per query" % (int(bc[-1]), int(bc[-1]) / (131072 * B), prof["n_candidates"]))
