import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev)
B = int(sys.argv[1])
q = torch.as_tensor(syn.rolling_queries(B, 20)).to(dev)
ws = _native.Workspace(dev)
for _ in range(40): out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws)
torch.cuda.synchronize()
