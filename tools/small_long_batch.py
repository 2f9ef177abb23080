"""Two / three queries with a long window that do not ride one pass of the three launches (W > 97 / 145): the batched long-window scan
against the loop of steps (PSH_FLAG_LONG_LOOP), per call."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from shadowing_amd import _native as N, synthetic as syn
dev = torch.device("cuda:0")
ds = torch.as_tensor(syn.dataset(32768, 4096, 2024)[:, 0, :].copy()).to(dev)
ws = N.Workspace(dev)
def timed(fn, steps=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for W in (100, 126, 150, 200, 252):
    for B in (2, 3):
        q = torch.as_tensor(syn.rolling_queries(B, W, 2025)).to(dev)
        info = {}
        d0, i0, s0 = N.scan_topk(ds, q, 1024, h=0, workspace=ws, info=info)
        d1, i1, s1 = N.scan_topk(ds, q, 1024, h=0, workspace=ws, flags=N.FLAG_LONG_LOOP)
        torch.cuda.synchronize()
        same = bool((s0 == 0).all()) and bool((s1 == 0).all()) and torch.equal(d0, d1) and torch.equal(i0, i1)
        a = timed(lambda: N.scan_topk(ds, q, 1024, h=0, workspace=ws))
        b = timed(lambda: N.scan_topk(ds, q, 1024, h=0, workspace=ws, flags=N.FLAG_LONG_LOOP))
        print(f"W={W} B={B} path {info.get('path')}: call {a:.3f} ms, loop of steps {b:.3f} ms, same {same}", flush=True)
