"""Probe: step time right after the device sat idle (what a short benchmark run sees) against the same run repeated at once."""
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
def mk(ns):
    return ([torch.cuda.Stream(dev) for _ in range(ns)], [_native.Workspace(dev) for _ in range(ns)],
            [(torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1, k, 2), dtype=torch.int32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev)) for _ in range(ns)])
def run(ctx, n, flags):
    streams, wss, outs = ctx
    ns = len(streams)
    t0 = time.perf_counter()
    for i in range(n):
        s = i % ns
        with torch.cuda.stream(streams[s]):
            _native.scan_topk(ds, q, k, h=h, workspace=wss[s], flags=flags, out=outs[s])
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
c3, c1 = mk(3), mk(1)
run(c3, 10, _native.FLAG_OVERLAP); run(c1, 10, 0)
for idle in (2.0, 0.5, 0.05):
    for name, ctx, fl in (("overlap x3", c3, _native.FLAG_OVERLAP), ("fused x1", c1, 0)):
        time.sleep(idle)
        w = run(ctx, 5, fl)
        a = run(ctx, 20, fl); b = run(ctx, 20, fl); c = run(ctx, 200, fl); d = run(ctx, 200, fl)
        print(f"idle {idle:4.2f} s, {name:10s}: warm-up 5 steps {w:7.1f} | 20 steps {a:7.1f} | 20 again {b:7.1f} | 200 {c:7.1f} | 200 again {d:7.1f} us/step", flush=True)
