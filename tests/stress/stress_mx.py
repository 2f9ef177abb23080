"""Adversarial parity stress of the matrix-core filter (W = 20, single query, sampled path):
outliers the bootstrap sample never saw, queries far smaller / larger than the data, extreme
fp32 scales, planted near-matches, heavy tails.  HIP scan vs the CPU oracle, bit for bit."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import oracle
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ws = _native.Workspace(dev)
bad = 0
t_all = time.time()
for case in range(n_cases):
    R = int(rng.choice([3000, 5000, 8192, 12000])); T = int(rng.choice([1024, 1500, 2048, 4096]))
    h = int(rng.integers(0, 30)); k = int(rng.choice([1, 64, 1024, 4000]))
    seed = int(rng.integers(1 << 30))
    ds = syn.dataset(R, T, seed)[:, 0, :].copy(); q = syn.gbm_log_returns((1, 20), seed + 1)
    kind = case % 10
    note = ""
    if kind == 0:   # spikes in a few rows (most of them unsampled)
        for r in rng.integers(0, R, 6): ds[r, rng.integers(0, T, 5)] *= float(10.0 ** rng.integers(2, 7))
        note = "spikes"
    elif kind == 1: # tiny query
        q *= float(10.0 ** -rng.integers(2, 6)); note = "tiny query"
    elif kind == 2: # huge query
        q *= float(10.0 ** rng.integers(2, 5)); note = "huge query"
    elif kind == 3: # extreme common scale
        s = float(10.0 ** rng.integers(-15, 15)); ds *= s; q *= s; note = f"scale {s:g}"
    elif kind == 4: # planted near-matches of the query
        for r in rng.integers(0, R, 50):
            t = int(rng.integers(0, T - 20)); ds[r, t:t + 20] = q[0] * (1 + 1e-3 * rng.standard_normal(20).astype(np.float32))
        ds[7, 100:120] = q[0]; note = "planted matches"
    elif kind == 5: # heavy tails
        ds = (0.01 * rng.standard_t(2.5, size=ds.shape)).astype(np.float32); note = "student-t(2.5)"
    elif kind == 6: # zero and constant rows, zeros inside the query
        ds[::7] = 0; ds[3::11] = 0.01; q[0, ::3] = 0; note = "zero/constant rows"
    elif kind == 7: # a whole block of rows 1000x louder than the rest
        ds[R // 3: R // 3 + 40] *= 1000.0; note = "loud rows"
    elif kind == 8: # quiet ensemble, one loud unsampled row
        ds *= 1e-3; ds[1] *= 1e5; note = "quiet + one loud row"
    else:
        note = "plain"
    ds_t = torch.as_tensor(ds).to(dev); q_t = torch.as_tensor(q).to(dev)
    d, idx, st, prof = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws, profile=True)
    torch.cuda.synchronize()
    ovf = int(st[0]) != 0
    if ovf:
        d, idx, _ = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws, exhaustive=True)
    od, oidx = oracle.scan_topk(ds, q, k, h=h)
    okd = np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32)); oki = np.array_equal(idx.cpu().numpy(), oidx)
    bad += not (okd and oki)
    print(f"{note:22s} R={R} T={T} h={h} k={k} path={prof['path']} cand={prof['n_candidates']} overflow={ovf} "
          f"d={okd} idx={oki}{'' if okd and oki else '   <<<<<< MISMATCH'}", flush=True)
print("mismatches:", bad, "time", round(time.time() - t_all, 1))
sys.exit(1 if bad else 0)
