"""Step time of the fused launch against the share of the even blocks of a pair (-DPSH_TUNING build: PSH_XCD_SKEW)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from shadowing_amd import _build
os.environ["PSH_LIB"] = str(_build.build(tuning=True))
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev); q = torch.as_tensor(syn.single_query(20, 1)[None]).to(dev)
ws = _native.Workspace(dev)
ref = None
for rnd in range(3):
    for sk in [int(a) for a in sys.argv[1:]] or [0, 3, 5, 8, 12]:
        os.environ["PSH_XCD_SKEW"] = str(sk)
        for _ in range(20): out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        res = (out[0].cpu().numpy().tobytes(), out[1].cpu().numpy().tobytes())
        if ref is None: ref = res
        print(f"skew {sk:3d}: {e0.elapsed_time(e1) / 300 * 1e3:7.2f} us/step  status {int(out[2][0])}  same {res == ref}", flush=True)
