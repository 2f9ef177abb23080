"""Probe: where one blocking PathShadowing.shadow(cuda=True) call (one Identity query, configs[1]) spends its time on the host:
argument handling / enqueue (PreparedShadow.launch) / waiting for the device / copying the results out of the pinned buffer."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import shadowing_amd as sa
from shadowing_amd import synthetic as syn
from shadowing_amd.path_shadowing import _torch, _dim_array
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)).cuda()
obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=20))
qs = [syn.gbm_log_returns((20,), 100 + i) for i in range(300)]
for q in qs[:20]: obj.shadow(q, k=1024, cuda=True)
slot = obj._sync_slot[1]
cur = torch.cuda.current_stream()
T = np.zeros(5)
for q in qs:
    t0 = time.perf_counter()
    x = _torch(_dim_array(q)); y = obj._dataset_tensor(); kind = obj._native_kind(x, y, 1024)
    dsr = obj._resident_dataset(y, ds.device); rows = obj._scan_rows_of(dsr)
    t1 = time.perf_counter()
    slot.launch(cur, x[:, 0, :])
    t2 = time.perf_counter()
    slot.event.synchronize()
    t3 = time.perf_counter()
    out = tuple(t.numpy().copy() for t in slot.host)
    t4 = time.perf_counter()
    T[:4] += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
t0 = time.perf_counter()
for q in qs: obj.shadow(q, k=1024, cuda=True)
T[4] = time.perf_counter() - t0
print("us per call: arguments %.1f, enqueue %.1f, wait %.1f, copy out %.1f; shadow() itself %.1f" % tuple(1e6 * T / len(qs)))
