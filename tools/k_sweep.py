"""Probe: one Identity(20) query at configs[1]'s ensemble for k = 1024 ... 16384 (testing.ipynb's own example call asks for k = 8192):
scan time per call with the stage times of one profiled call, and PathShadowing.shadow() with its gathered paths."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import shadowing_amd as sa
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)).cuda()
q = torch.as_tensor(syn.single_query(20, 1)[None]).cuda()
ws = _native.Workspace(dev)
for k in (1024, 4096, 8192, 16384):
    for _ in range(3): out = _native.scan_topk(ds[:, 0, :], q, k, h=20, workspace=ws)
    torch.cuda.synchronize()
    *_, prof = _native.scan_topk(ds[:, 0, :], q, k, h=20, workspace=ws, profile=True)
    t0 = time.perf_counter()
    for _ in range(50): _native.scan_topk(ds[:, 0, :], q, k, h=20, workspace=ws)
    torch.cuda.synchronize()
    print(k, "us per call", round((time.perf_counter() - t0) / 50 * 1e6, 1), {a: round(b, 4) if isinstance(b, float) else b for a, b in prof.items()})
obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=20))
x = syn.single_query(20, 1)
for k in (1024, 8192):
    for _ in range(10): obj.shadow(x, k=k, cuda=True)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); obj.shadow(x, k=k, cuda=True); ts.append(time.perf_counter() - t0)
    print("shadow() k", k, "us per call", round(float(np.median(ts)) * 1e6, 1), "(median of 50)")
