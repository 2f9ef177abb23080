import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from shadowing_amd import _build
os.environ["PSH_LIB"] = str(_build.build(tuning=True))   # the instrumented build (-DPSH_TUNING): env overrides, time stamps
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev); q = torch.as_tensor(syn.single_query(20, 1)[None]).to(dev)
ws = _native.Workspace(dev)
buf = torch.zeros(64, dtype=torch.int64, device=dev)
os.environ["PSH_DBG_SELECT_PTR"] = str(buf.data_ptr())
for rep in range(3):
    *_, prof = _native.scan_topk(ds, q, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, h=20, workspace=ws, profile=True)
t = buf.cpu().numpy()[:8].astype(np.float64) * 0.01  # us
names = ["slice prefix", "load keys", "min/max + radix select", "assign slots", "fetch (r,t)", "sort", "write out"]
print("select_ms", prof["select_ms"], "candidates", prof["n_candidates"])
for i, n in enumerate(names):
    print(f"  {n:14s} {t[i+1]-t[i]:7.2f} us")
print(f"  total in-kernel {t[7]-t[0]:.2f} us")

