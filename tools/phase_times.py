import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
os.environ["PSH_LIB"] = str(REPO / "shadowing_amd/lib/libpsh_hip_phase.so")
sys.path.insert(0, str(REPO))
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev); q = torch.as_tensor(syn.single_query(20, 1)[None]).to(dev)
ws = _native.Workspace(dev)
buf = torch.zeros(2 * 8192 + 5 * 8192, dtype=torch.int64, device=dev)
os.environ["PSH_DBG_TIMES_PTR"] = str(buf.data_ptr())
for rep in range(3):
    *_, prof = _native.scan_topk(ds, q, 1024, h=20, workspace=ws, profile=True)
nw = prof["grid_blocks"] * 16
ph = buf.cpu().numpy()[2 * 8192:2 * 8192 + 5 * nw].reshape(nw, 5).astype(np.float64)
units = ph[:, 4]
print("scan_ms", prof["scan_ms"], "waves", nw, "units/wave mean", units.mean())
names = ["wait prefetched loads (vmcnt)", "ds_write tile + fence", "flush + grab + prefetch issue", "compute (ds_read, x load, VALU, rare path)"]
tot = ph[:, :4].sum(1).mean()
for i, n in enumerate(names):
    print(f"  {n:45s}: {ph[:, i].sum() / units.sum():8.0f} cycles/unit  ({100 * ph[:, i].mean() / tot:.1f}%)")
print(f"  total {tot / units.mean():.0f} cycles per unit per wave (s_memtime ticks)")
