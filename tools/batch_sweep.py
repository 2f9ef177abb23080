import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev)
for B in (1, 2, 3, 4, 8, 16, 32, 64, 128):
    q = torch.as_tensor(syn.rolling_queries(B, 20)).to(dev)
    ws = _native.Workspace(dev)
    for _ in range(5): out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    print(B, round(e0.elapsed_time(e1) / 30 * 1e3, 1), "us/call", round(e0.elapsed_time(e1) / 30 * 1e3 / B, 1), "us/query", "status", int(out[2].max()))
