"""Probe: does the 3-stream overlap steady state depend on run length / on the event pairs bench.py records?  With
TRACE=1 it runs 8 timed regions of 300 steps and prints their wall-clock windows (for tools/overlap_timeline.py)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
R, T, W, h, k = 32768, 4096, 20, 20, 1024
ds = torch.from_numpy(syn.dataset(R, T, seed=0)).to(dev)[:, 0, :]
q = torch.from_numpy(syn.single_query(W, syn.QUERY_SEED)[None, :].copy()).to(dev)
_native.load()
NS = int(os.environ.get("NSTREAMS", "3"))
def run(ns, steps, flags, events=False):
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    wss = [_native.Workspace(dev) for _ in range(ns)]
    outs = [(torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1, k, 2), dtype=torch.int32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev)) for _ in range(ns)]
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps // 4 + 1)]
    for a, b in evs: a.record(); b.record()
    def go(n, ev=False):
        for i in range(n):
            s = i % ns
            with torch.cuda.stream(streams[s]):
                _native.scan_topk(ds, q, k, h=h, workspace=wss[s], flags=flags, out=outs[s], scan_events=evs[i // 4] if (ev and i % 4 == 0) else None)
    go(30); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(steps, events); torch.cuda.synchronize(); el = time.perf_counter() - t0
    return 1e6 * el / steps
if os.environ.get("TRACE"):
    for rep in range(10):
        print(f"region {rep}: {run(NS, 300, _native.FLAG_OVERLAP, rep % 2 == 1):7.2f} us/step", flush=True)
else:
    for steps in (20, 100, 300, 1000, 3000):
        print(steps, "steps:", " ".join(f"{run(NS, steps, _native.FLAG_OVERLAP):7.2f}" for _ in range(3)), "| with events every 4th:", " ".join(f"{run(NS, steps, _native.FLAG_OVERLAP, True):7.2f}" for _ in range(3)), flush=True)
