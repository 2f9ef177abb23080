"""Phase boundaries of the fused single-launch scan, per block (wave 0), from the device's wall clock
(-DPSH_TUNING build).  Usage: python tools/fused_times.py [boot_units]"""
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
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
hist = []
for rep in range(12):
    ev[0].record()
    d, idx, st = _native.scan_topk(ds, q, 1024, h=20, workspace=ws)
    ev[1].record()
    torch.cuda.synchronize()
    tt = buf.cpu().numpy().reshape(256, 8).astype(np.float64) * 0.01
    hist.append([float(np.median(tt[x::8, 4] - tt[:, 0].min())) for x in range(8)])
print("end of the scan per XCD (median of its blocks, us), launch after launch:")
for h_ in hist[2:]:
    print("  " + " ".join(f"{v:6.1f}" for v in h_))
print("status", int(st[0]), "call", ev[0].elapsed_time(ev[1]) * 1e3, "us")
t = buf.cpu().numpy().reshape(256, 8).astype(np.float64) * 0.01     # us
t0 = t[:, 0].min()
names = ["start", "A done (bootstrap)", "sweep done (barrier 1)", "B done (tau2)", "scan done", "block barrier", "counts swept (barrier 2)", "ranked+written"]
for i, n in enumerate(names):
    c = t[:, i] - t0
    print(f"{n:28s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
d = np.diff(t, axis=1)
for i, n in enumerate(names[1:]):
    print(f"  phase -> {n:26s} median {np.median(d[:, i]):7.2f}  max {d[:, i].max():7.2f} us")
# where the scan's tail comes from: end of the scan phase per XCD (blocks are dealt round-robin: XCD = block % 8)
c = t[:, 4] - t0
print("scan done per XCD (block % 8): median / max")
for x in range(8):
    print(f"  xcd {x}: {np.median(c[x::8]):7.2f} {c[x::8].max():7.2f}   start of scan (B done) median {np.median(t[x::8, 3] - t0):6.2f}")
print("ten slowest blocks:", np.argsort(-c)[:10].tolist(), np.sort(-c)[:10].round(1).tolist())
