import os, sys, subprocess
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from shadowing_amd import _build
os.environ["PSH_LIB"] = str(_build.build(tuning=True))   # the instrumented build (-DPSH_TUNING): env overrides, time stamps
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev); q = torch.as_tensor(syn.single_query(20, 1)[None]).to(dev)
ws = _native.Workspace(dev)
for bpc in (2, 3, 4, 5, 6, 8):
    os.environ["PSH_BLOCKS_PER_CU"] = str(bpc)
    ts = []
    for _ in range(6):
        *_, prof = _native.scan_topk(ds, q, 1024, h=20, workspace=ws, profile=True)
        ts.append(prof["scan_ms"])
    print(bpc, prof["grid_blocks"], "scan_ms min %.4f med %.4f" % (min(ts), sorted(ts)[3]), "boot %.4f thr %.4f sel %.4f" % (prof["sample_ms"], prof["threshold_ms"], prof["select_ms"]), flush=True)
