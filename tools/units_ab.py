"""A/B in one process: sample size of the overlap launches (PSH_STREAM_UNITS, tuning build), alternating runs of 1000 steps."""
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
NS = 3
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
wss = [_native.Workspace(dev) for _ in range(NS)]
outs = [(torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1, k, 2), dtype=torch.int32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev)) for _ in range(NS)]
tot = torch.zeros(1, dtype=torch.int32, device=dev)
def run(n):
    t0 = time.perf_counter()
    for i in range(n):
        s = i % NS
        with torch.cuda.stream(streams[s]):
            _native.scan_topk(ds, q, k, h=h, workspace=wss[s], flags=_native.FLAG_OVERLAP, out=outs[s])
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
run(300)
res = {}
for rep in range(4):
    for units in sys.argv[1:] or ("2048", "1024", "512"):
        os.environ["PSH_STREAM_UNITS"] = units
        run(100)
        res.setdefault(units, []).append(run(1000))
        assert int(outs[0][2].item()) == 0
for u, v in res.items():
    print(f"units {u}: " + " ".join(f"{x:6.2f}" for x in v) + f"  mean {np.mean(v):6.2f} us/step")
