"""The fused launch on THIS box: how far apart the XCDs end their shares, and the step time against the static even/odd
shares (PSH_XCD_SKEW) -- -DPSH_TUNING build.  (The stealable-tail variant DESIGN.md reports was measured with this script
and a patched kernel that read PSH_TAIL256; the patch is not in the tree.)"""
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
buf = torch.zeros(8 * 256, dtype=torch.int64, device=dev)
os.environ["PSH_DBG_TIMES_PTR"] = str(buf.data_ptr())
ref = None
for rnd in range(2):
    for tl, sk in ((0, 0), (0, 6), (0, 12)):
        os.environ["PSH_TAIL256"] = str(tl); os.environ["PSH_XCD_SKEW"] = str(sk)
        for _ in range(20): out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(256, 8).astype(np.float64) * 0.01
        c = t[:, 4] - t[:, 0].min()
        xm = [float(np.median(c[x::8])) for x in range(8)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        res = (out[0].cpu().numpy().tobytes(), out[1].cpu().numpy().tobytes())
        if ref is None: ref = res
        print(f"tail {tl:3d} skew {sk}: {e0.elapsed_time(e1) / 300 * 1e3:7.2f} us/step  same {res == ref} status {int(out[2][0])}  scan end per XCD "
              + " ".join(f"{v:5.1f}" for v in xm) + f"  spread {max(xm) - min(xm):4.1f} slowest block {c.max():5.1f}", flush=True)
