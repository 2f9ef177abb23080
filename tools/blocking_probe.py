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
fs = obj._fast
raw = torch._C._cuda_getCurrentRawStream(ds.device.index)
T = np.zeros(4)
for q in qs:
    t0 = time.perf_counter()
    ok = obj._shadow_fast.__func__ is not None and (1024 == fs["k"] and obj.dataset is fs["owner"])   # (stand-in for the checks)
    t1 = time.perf_counter()
    st, res = slot.call(raw, q, None)
    t2 = time.perf_counter()
    T[:2] += (t1 - t0, t2 - t1)
t0 = time.perf_counter()
for q in qs: obj.shadow(q, k=1024, cuda=True)
T[2] = time.perf_counter() - t0
ts = []
for q in qs:
    t0 = time.perf_counter(); r = obj.shadow(q, k=1024, cuda=True); ts.append(1e6 * (time.perf_counter() - t0))
print("us per call: library call (psh_shadow_blocking through ctypes, results as views) %.1f; shadow() itself mean %.1f, median %.1f, p90 %.1f; fused launch served: %s"
      % (1e6 * T[1] / len(qs), 1e6 * T[2] / len(qs), np.median(ts), np.percentile(ts, 90), slot.last_fused))
# results kept by the caller: every call takes a fresh block until the pool is used up, then results are copied out
kept = []
ts = []
for q in qs[:40]:
    t0 = time.perf_counter(); kept.append(obj.shadow(q, k=1024, cuda=True)); ts.append(1e6 * (time.perf_counter() - t0))
print("caller keeps every result: median %.1f us per call (first calls allocate pinned blocks: max %.0f)" % (np.median(ts[12:]), max(ts)))
del kept
for kk in (4096, 8192):
    for q in qs[:10]: obj.shadow(q, k=kk, cuda=True)
    ts = []
    for q in qs[:100]:
        t0 = time.perf_counter(); r = obj.shadow(q, k=kk, cuda=True); ts.append(1e6 * (time.perf_counter() - t0))
    print("k = %d: shadow() median %.1f, p90 %.1f us; fused launch served: %s" % (kk, np.median(ts), np.percentile(ts, 90), obj._sync_slot[1].last_fused))
for q in qs[:5]: obj.shadow(q, k=1024, cuda=True)

# ---- the same loop with admission hints (PathShadowing(hint="auto"), psh_profile.tau_hint): ROLLING query dates -- what the
#      hint is for -- and unrelated queries (where it mostly falls short and backs off), each against the same object without
def loop(o, queries):
    for q in queries[:20]:
        o.shadow(q, k=1024, cuda=True)
    seen = {"ok": 0, "short": 0, None: 0}
    ts = []
    for q in queries:
        t0 = time.perf_counter()
        o.shadow(q, k=1024, cuda=True)
        ts.append(1e6 * (time.perf_counter() - t0))
        seen[o.last_hint] += 1
    return (float(np.median(ts)), float(np.mean(ts))), seen
hinted = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=20), hint="auto")
rolling = list(syn.rolling_queries(300, 20, 7))
for name, queries in (("rolling dates", rolling), ("unrelated queries", qs)):
    us_plain, _ = loop(obj, queries)
    us_hint, seen = loop(hinted, queries)
    print("%s: shadow() median %.1f / mean %.1f us per call without hints, median %.1f / mean %.1f with hint=\"auto\" (hints held %d, fell short %d, calls without one %d)"
          % (name, us_plain[0], us_plain[1], us_hint[0], us_hint[1], seen["ok"], seen["short"], seen[None]))

# ---- where a hinted call's time goes: the slot API directly, the same query, hint None vs its own k-th acc x 1.1
q = qs[0]
d0, _, _ = obj.shadow(q, k=1024, cuda=True)
xn2 = float(np.asarray(q, np.float64) @ np.asarray(q, np.float64))
good = float(d0[0, -1]) ** 2 * xn2 * 1.1
slot = obj._sync_slot[1]
for name, hv in (("no hint", None), ("hint", good), ("no hint", None), ("hint", good)):
    tw = 0.0
    for _ in range(200):
        t0 = time.perf_counter(); st, res = slot.call(raw, q, hv); t1 = time.perf_counter()
        tw += t1 - t0
    print("%-8s library call %.1f us, status %d" % (name, 1e6 * tw / 200, st))

# ---- per-call times of the hinted object over the 300 rolling dates
times = []
for q in rolling:
    t0 = time.perf_counter(); hinted.shadow(q, k=1024, cuda=True); t = time.perf_counter() - t0
    times.append((1e6 * t, hinted.last_hint, hinted.last_path))
ts = np.array([t for t, _, _ in times])
print("per call: median %.1f, mean %.1f, p90 %.1f, max %.1f us" % (np.median(ts), ts.mean(), np.percentile(ts, 90), ts.max()))
for i in np.argsort(-ts)[:12]:
    print("  call %d: %.0f us, hint %s, path %s" % (i, times[i][0], times[i][1], times[i][2]))
