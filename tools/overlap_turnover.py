"""Tuning-build probe: S-only steady state on 2 streams with per-launch block stamps -> how long a CU sits between the
end of a block of launch i and the start of a block of launch i+1 (k-th end paired with k-th start), and the phases
inside a block.  PSH_LIB must point at libpsh_hip_tuning.so."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn

dev = torch.device("cuda", 0)
R, T, W, h, k = 32768, 4096, 20, 20, 1024
ds = torch.from_numpy(syn.dataset(R, T, seed=0)).to(dev)[:, 0, :]
q = torch.from_numpy(syn.single_query(W, syn.QUERY_SEED)[None, :].copy()).to(dev)
_native.load()
FL = _native.FLAG_OVERLAP
NS = int(os.environ.get("NSTREAMS", "2"))
NB = 512
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
wss = [_native.Workspace(dev) for _ in range(NS)]
os.environ["PSH_STREAM_SKIP"] = "0"
for s in range(NS):
    with torch.cuda.stream(streams[s]):
        _native.scan_topk(ds, q, k, h=h, workspace=wss[s], flags=FL)
torch.cuda.synchronize()
os.environ["PSH_STREAM_SKIP"] = os.environ.get("SKIP", "3")
L = 12
bufs = [torch.zeros(NB * 8, dtype=torch.int64, device=dev) for _ in range(L)]
for i in range(20 + L):
    s = i % NS
    if i >= 20:
        os.environ["PSH_DBG_TIMES_PTR"] = hex(bufs[i - 20].data_ptr())
    with torch.cuda.stream(streams[s]):
        _native.scan_topk(ds, q, k, h=h, workspace=wss[s], flags=FL)
torch.cuda.synchronize()
os.environ.pop("PSH_DBG_TIMES_PTR"); os.environ["PSH_STREAM_SKIP"] = "0"
t = [b.cpu().numpy().reshape(NB, 8).astype(np.float64) / 100.0 for b in bufs]
t = [x[x[:, 0] > 0] for x in t]
t0 = t[0][:, 0].min()
for i, x in enumerate(t):
    st, rdy, first, done, pub = (x[:, j] - t0 for j in (0, 1, 4, 2, 3))
    print(f"launch {i}: blocks {len(x)} start {st.min():7.1f}..{st.max():7.1f} (med {np.median(st):7.1f})  end {pub.min():7.1f}..{pub.max():7.1f} (med {np.median(pub):7.1f})"
          f" | setup {np.median(rdy - st):.2f} first-data {np.median(first - st):.2f} scan {np.median(done - first):.2f} tail {np.median(pub - done):.2f} block {np.median(pub - st):.2f}")
    if i:
        pe = np.sort(t[i - 1][:, 3] - t0); ns = np.sort(st)
        n = min(len(pe), len(ns)); gap = ns[:n] - pe[:n]
        print(f"      turnover (k-th start of this launch - k-th end of the previous): median {np.median(gap):.2f} p10 {np.percentile(gap, 10):.2f} p90 {np.percentile(gap, 90):.2f} us;"
              f" launch-to-launch median start {np.median(st) - np.median(t[i - 1][:, 0] - t0):.2f} us")
