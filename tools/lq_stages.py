"""Probe: stage times of a batch of long-window queries (the batched long-window scan, psh_lq.hip, through the separate launches:
sample / threshold / scan / select by HIP events) on configs[1]'s ensemble."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from shadowing_amd import _native as N, synthetic as syn
dev = torch.device("cuda:0")
ds = torch.as_tensor(syn.dataset(32768, 4096, 2024)[:, 0, :].copy()).to(dev)
ws = N.Workspace(dev)
for W in [int(a) for a in (sys.argv[1:2] or ["126"])[0].split(",")]:
    for B in (16, 64, 512):
        q = torch.as_tensor(syn.rolling_queries(B, W, 2025)).to(dev)
        N.scan_topk(ds, q, 1024, h=0, workspace=ws)
        torch.cuda.synchronize()
        _, _, st, prof = N.scan_topk(ds, q, 1024, h=0, workspace=ws, profile=True)
        print(json.dumps(dict(W=W, B=B, status_max=int(st.max().item()), **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in prof.items()})), flush=True)
