"""Probe: steps of configs[1] issued alternately on S streams (one workspace per stream): fused launch, separate launches,
the three overlap-friendly launches (PSH_FLAG_OVERLAP).  Checks the first result of every mode against the reference's
golden vector, then times steps.  (tools; not product code)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn

dev = torch.device("cuda", 0)
R, T, W, h, k = 32768, 4096, 20, 20, 1024
ds = torch.from_numpy(syn.dataset(R, T, seed=0)).to(dev)[:, 0, :]
q = torch.from_numpy(syn.single_query(W, syn.QUERY_SEED)[None, :].copy()).to(dev)
_native.load()
g = np.load(Path(__file__).resolve().parents[1] / "tests/golden/cfg2_R32768.npz")
gd = np.sort(g["d"], axis=1)

def check(flags):
    ws = _native.Workspace(dev)
    info = {}
    d, idx, st = _native.scan_topk(ds, q, k, h=h, workspace=ws, flags=flags, info=info)
    torch.cuda.synchronize()
    ok = (int(st.max().item()) == 0 and np.array_equal(d.cpu().numpy().view(np.uint32), gd.view(np.uint32))
          and {tuple(v) for v in idx.cpu().numpy()[0]} == {tuple(v) for v in g["idx"][0]})
    return ok, info.get("path"), int(st.max().item())

def run(nstreams, flags, steps=300, warm=30):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    wss = [_native.Workspace(dev) for _ in range(nstreams)]
    outs = [(torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1, k, 2), dtype=torch.int32, device=dev)) for _ in range(nstreams)]
    sts = []
    def go(n):
        for i in range(n):
            s = i % nstreams
            with torch.cuda.stream(streams[s]):
                d, idx, st = _native.scan_topk(ds, q, k, h=h, workspace=wss[s], flags=flags, out=outs[s])
                sts.append(st)
    go(warm); torch.cuda.synchronize(); sts.clear()
    t0 = time.perf_counter(); go(steps); th = time.perf_counter() - t0; torch.cuda.synchronize(); el = time.perf_counter() - t0
    bad = int(torch.stack(sts).max().item())
    okk = all(np.array_equal(o[0].cpu().numpy().view(np.uint32), gd.view(np.uint32)) for o in outs)
    return 1e6 * el / steps, 1e6 * th / steps, bad, okk

modes = (("fused", 0), ("separate", _native.FLAG_NO_FUSE), ("overlap", _native.FLAG_OVERLAP))
only = sys.argv[1:] or [m[0] for m in modes]
for name, fl in modes:
    if name not in only:
        continue
    print(name, "first result == golden, path, status:", check(fl), flush=True)
    for ns in (1, 2, 3, 4):
        if name == "fused" and ns > 2:
            continue
        for rep in range(2):
            us, host, bad, okk = run(ns, fl)
            print(f"{name:9s} streams={ns} step {us:7.2f} us  host enqueue {host:6.2f} us  worst status {bad}  results==golden {okk}", flush=True)
