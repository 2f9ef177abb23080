"""Probe: PathShadowing.shadow_async() against blocking shadow(cuda=True) calls at configs[1] (numpy query in, numpy triple
out, the 1024 gathered paths included)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import shadowing_amd as sa
from shadowing_amd import synthetic as syn
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)).cuda()
obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=20))
qs = [syn.gbm_log_returns((20,), 100 + i) for i in range(200)]
for q in qs[:10]: obj.shadow(q, k=1024, cuda=True)
t0 = time.perf_counter()
for q in qs: obj.shadow(q, k=1024, cuda=True)
t_block = (time.perf_counter() - t0) / len(qs)
def windowed(depth):
    """at most `depth` calls outstanding: the result of the oldest is taken when the window is full (slots are reused)"""
    out, pend = [], []
    t0 = time.perf_counter()
    for q in qs:
        pend.append(obj.shadow_async(q, k=1024))
        if len(pend) == depth:
            out.append(pend.pop(0).result())
    out += [h.result() for h in pend]
    return (time.perf_counter() - t0) / len(qs), out
windowed(6)
for depth in (1, 2, 3, 6, 12):
    t, out = windowed(depth)
    print(f"shadow_async, {depth:2d} calls outstanding: {1e6 * t:6.1f} us per call (results on the host)")
print(f"blocking shadow(cuda=True): {1e6 * t_block:.1f} us per call")
ref = [obj.shadow(q, k=1024, cuda=True) for q in qs[:20]]
assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1]) for a, b in zip(out[:20], ref))
