"""Is the step host-bound?  Host enqueue time per scan_topk call against the GPU time per step (events around the loop)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev); q = torch.as_tensor(syn.single_query(20, 1)[None]).to(dev)
ws = _native.Workspace(dev)
N = 300
for flags, name in ((0, "fused"), (_native.FLAG_NO_FUSE, "separate launches")):
    for variant in ("plain", "out="):
        out = (torch.empty((1, 1024), dtype=torch.float32, device=dev), torch.empty((1, 1024, 2), dtype=torch.int32, device=dev)) if variant == "out=" else None
        for _ in range(20):
            _native.scan_topk(ds, q, 1024, h=20, workspace=ws, flags=flags, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(N):
            _native.scan_topk(ds, q, 1024, h=20, workspace=ws, flags=flags, out=out)
        e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{name:18s} {variant:6s}: host enqueue {1e6*(t1-t0)/N:7.1f} us/call, GPU {1e3*e0.elapsed_time(e1)/N:7.1f} us/step, wall {1e6*(t2-t0)/N:7.1f} us/step")
