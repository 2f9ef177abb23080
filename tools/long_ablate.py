"""Tuning build only: the long-window scan (stream_scan_long_kernel) with parts switched off (PSH_DBG: 4 no survivor handling,
8 one K-step of the band; results invalid) -- one query per call on one stream, and the scan kernel's own time by events."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shadowing_amd import _native as N, synthetic as syn
dev = torch.device("cuda:0")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 126
ds = torch.as_tensor(syn.dataset(32768, 4096, 2024)[:, 0, :].copy()).to(dev)
q = torch.as_tensor(syn.rolling_queries(1, W, 2025)).to(dev)
ws = N.Workspace(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record()
for _ in range(5):
    N.scan_topk(ds, q, 1024, h=0, workspace=ws, scan_events=(e0, e1))
torch.cuda.synchronize()
ts = []
for _ in range(30):
    N.scan_topk(ds, q, 1024, h=0, workspace=ws, scan_events=(e0, e1))
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("PSH_DBG=%s W=%d scan kernel: median %.1f us, min %.1f us" % (os.environ.get("PSH_DBG", "0"), W, ts[len(ts) // 2], ts[0]))
