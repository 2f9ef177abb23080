"""Probe: the library's own split of a blocking shadow() call (psh_shadow_blocking: us enqueueing / us waiting) beside the
caller's clock, configs[1]."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import shadowing_amd as sa
from shadowing_amd import synthetic as syn
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)).cuda()
obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=20))
qs = [syn.gbm_log_returns((20,), 100 + i) for i in range(300)]
for q in qs[:20]: obj.shadow(q, k=1024, cuda=True)
slot = obj._sync_slot[1]
raw = torch._C._cuda_getCurrentRawStream(ds.device.index)
en, wa, lib, tot, sl, fd = [], [], [], [], [], []
for q in qs:
    t0 = time.perf_counter(); st, res = slot.call(raw, q, None); t1 = time.perf_counter()
    lib.append(1e6 * (t1 - t0))
    del res
    en.append(float(slot.blocks[0].times[0])); wa.append(float(slot.blocks[0].times[1])); sl.append(float(slot.blocks[0].times[2])); fd.append(float(slot.blocks[0].times[3]))
for q in qs:
    t0 = time.perf_counter(); r = obj.shadow(q, k=1024, cuda=True); tot.append(1e6 * (time.perf_counter() - t0)); del r
print("psh_shadow_blocking: enqueue %.1f us, wait %.1f us of which %.1f until the launch's first block ran, %.1f until the first completion word (library's clock); ctypes call %.1f us; shadow() %.1f us (medians)"
      % (np.median(en), np.median(wa), np.median(sl), np.median(fd), np.median(lib), np.median(tot)))

# ---- A/B on the same box: (1) psh_shadow_blocking, (2) round 5's slot (fused launch + gather launch writing the pinned buffer,
#      event wait), (3) the fused launch alone, results in HBM, launch after launch (what bench.py's one-stream leg times)
from shadowing_amd import _native
rows = obj._fast["rows"]; dsr = obj._fast["ds"]
old = _native.PreparedShadow(rows, dsr, 20, 1024, 20, _native.Workspace(ds.device), 0, host_direct=True)
cur = torch.cuda.current_stream()
xs = [torch.as_tensor(q)[None, :] for q in qs]
for x in xs[:20]:
    old.launch(cur, x); old.event.synchronize()
t_old = []
for x in xs:
    t0 = time.perf_counter(); old.launch(cur, x); old.event.synchronize(); t_old.append(1e6 * (time.perf_counter() - t0))
t_new = []
for q in qs:
    t0 = time.perf_counter(); st, res = slot.call(raw, q, None); t_new.append(1e6 * (time.perf_counter() - t0)); del res
qd = torch.as_tensor(np.stack(qs)).cuda()
ws2 = _native.Workspace(ds.device)
out = (torch.empty((1, 1024), dtype=torch.float32, device="cuda"), torch.empty((1, 1024, 2), dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"))
for i in range(20): _native.scan_topk(rows, qd[i:i + 1], 1024, h=20, workspace=ws2, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300): _native.scan_topk(rows, qd[i:i + 1], 1024, h=20, workspace=ws2, out=out)
torch.cuda.synchronize()
b2b = 1e6 * (time.perf_counter() - t0) / 300
t_one = []
for i in range(300):
    t0 = time.perf_counter(); _native.scan_topk(rows, qd[i:i + 1], 1024, h=20, workspace=ws2, out=out); torch.cuda.synchronize(); t_one.append(1e6 * (time.perf_counter() - t0))
print("same box: psh_shadow_blocking %.1f us; round 5's slot (fused + gather launches, event wait; results still to be copied out) %.1f us; "
      "fused launch alone, results in HBM: %.1f us per launch back to back, %.1f us launch + synchronize one at a time"
      % (np.median(t_new), np.median(t_old), b2b, np.median(t_one)))
