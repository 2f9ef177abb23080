import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from shadowing_amd import _build
os.environ["PSH_LIB"] = str(_build.build(tuning=True))
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev); q = torch.as_tensor(syn.single_query(20, 1)[None]).to(dev)
ws = _native.Workspace(dev)
buf = torch.zeros(64, dtype=torch.int64, device=dev)
os.environ["PSH_DBG_SELECT_PTR"] = str(buf.data_ptr())
for rep in range(5):
    out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws, flags=_native.FLAG_FILTER_VALU)
    torch.cuda.synchronize()
t = buf.cpu().numpy()[:8].astype(np.float64) * 0.01
print("rank kernel stamps (us):", (t - t[0]).round(2))
