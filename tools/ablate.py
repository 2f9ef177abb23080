import os, sys, subprocess
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
code = r'''
import sys; sys.path.insert(0, "%s")
import torch, numpy as np
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev); q = torch.as_tensor(syn.rolling_queries(int(sys.argv[1]), 20, 1)).to(dev)
ws = _native.Workspace(dev); ts = []
for _ in range(6):
    *_, prof = _native.scan_topk(ds, q, 1024, h=20, workspace=ws, profile=True); ts.append(prof["scan_ms"])
print("scan_ms min %%.4f" %% min(ts), "candidates", prof["n_candidates"])
''' % REPO
for lib in ("libpsh_hip.so", "libpsh_hip_abl1.so", "libpsh_hip_abl2.so"):
    for B in ("1", "8"):
        env = dict(os.environ, PSH_LIB=str(REPO / "shadowing_amd/lib" / lib))
        out = subprocess.run([sys.executable, "-c", code, B], env=env, capture_output=True, text=True)
        print(lib, "B=" + B, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
