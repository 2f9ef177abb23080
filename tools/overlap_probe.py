"""Tuning-build probe of the overlap-friendly launches: S-only steady state (P and R skipped after a complete warm-up
call per workspace), per-block time stamps of S.  PSH_LIB must point at libpsh_hip_tuning.so."""
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
FL = _native.FLAG_OVERLAP

def run(nstreams, skip, steps=300, warm=30):
    os.environ["PSH_STREAM_SKIP"] = "0"
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    wss = [_native.Workspace(dev) for _ in range(nstreams)]
    outs = [(torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1, k, 2), dtype=torch.int32, device=dev)) for _ in range(nstreams)]
    def go(n):
        for i in range(n):
            s = i % nstreams
            with torch.cuda.stream(streams[s]):
                _native.scan_topk(ds, q, k, h=h, workspace=wss[s], flags=FL, out=outs[s])
    go(nstreams); torch.cuda.synchronize()
    os.environ["PSH_STREAM_SKIP"] = str(skip)
    go(warm); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(steps); torch.cuda.synchronize(); el = time.perf_counter() - t0
    os.environ["PSH_STREAM_SKIP"] = "0"
    return 1e6 * el / steps

for skip, what in ((0, "P S R"), (2, "P S"), (1, "S R"), (3, "S only")):
    for ns in (1, 2, 3):
        print(f"{what:7s} streams={ns}: " + "  ".join(f"{run(ns, skip):7.2f}" for _ in range(3)), "us/step", flush=True)

# per-block stamps of S (single stream, S only): start -> ready -> scan done -> published
buf = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
os.environ["PSH_DBG_TIMES_PTR"] = hex(buf.data_ptr())
ws = _native.Workspace(dev)
_native.scan_topk(ds, q, k, h=h, workspace=ws, flags=FL); torch.cuda.synchronize()
for rep in range(3):
    buf.zero_(); torch.cuda.synchronize()
    _native.scan_topk(ds, q, k, h=h, workspace=ws, flags=FL); torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(256, 8).astype(np.float64) / 100.0     # us (100 MHz)
    t0 = t[:, 0].min()
    st, rdy, done, pub = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0, t[:, 3] - t0
    print(f"S blocks: start med {np.median(st):.2f} max {st.max():.2f} | ready-start med {np.median(rdy - st):.2f} max {(rdy - st).max():.2f} | "
          f"scan med {np.median(done - rdy):.2f} min {(done - rdy).min():.2f} max {(done - rdy).max():.2f} | end med {np.median(pub):.2f} max {pub.max():.2f} | publish med {np.median(pub - done):.2f}")
    print("   end per XCD (median):", " ".join(f"{np.median(pub[x::8]):.1f}" for x in range(8)))
os.environ.pop("PSH_DBG_TIMES_PTR")
