"""Randomised parity stress: HIP scan vs the CPU oracle over odd shapes (development aid)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import oracle
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ws = _native.Workspace(dev)
bad = 0
cases = [  # fixed, then random
    (2048, 4096, 20, 20, 8192, 2), (2048, 4096, 20, 20, 10000, 1), (3000, 2000, 20, 0, 16384, 1),
    (70000, 128, 20, 20, 500, 3), (9000, 1000, 24, 12, 300, 5), (9000, 1000, 17, 3, 300, 2), (9000, 1000, 32, 0, 300, 2),
    (5000, 1501, 20, 20, 200, 2), (513, 4099, 20, 1, 1024, 1), (20000, 60, 20, 20, 100, 2), (1200, 900, 16, 4, 64, 1100),
]
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    W = int(rng.choice([3, 8, 15, 16, 17, 20, 20, 20, 21, 31, 32, 33, 64, 100, 256]))
    h = int(rng.integers(0, 30))
    T = int(rng.integers(W + h + 1, 3000))
    R = int(rng.integers(1, 6000))
    N = R * (T - W - h + 1)
    k = int(min(N, rng.choice([1, 7, 64, 500, 1024, 3000])))
    B = int(rng.choice([1, 1, 2, 5, 17]))
    cases.append((R, T, W, h, k, B))
t_all = time.time()
for (R, T, W, h, k, B) in cases:
    seed = int(rng.integers(1 << 30))
    ds = syn.dataset(R, T, seed); q = syn.gbm_log_returns((B, W), seed + 1)
    if rng.random() < 0.3:      # plant near-duplicates / exact duplicates
        ds[R // 2:] = ds[: R - R // 2]
    ds_t = torch.as_tensor(ds[:, 0, :].copy()).to(dev); q_t = torch.as_tensor(q).to(dev)
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws)
    torch.cuda.synchronize()
    badq = torch.nonzero(st != 0).flatten()
    if badq.numel():
        d2, idx2, _ = _native.scan_topk(ds_t, q_t[badq].contiguous(), k, h=h, workspace=ws, exhaustive=True)
        d[badq] = d2; idx[badq] = idx2
    od, oidx = oracle.scan_topk(ds, q, k, h=h)
    okd = np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32)); oki = np.array_equal(idx.cpu().numpy(), oidx)
    flag = "" if (okd and oki) else "   <<<<<< MISMATCH"
    bad += not (okd and oki)
    print(f"R={R} T={T} W={W} h={h} k={k} B={B} overflowed={badq.numel()} d={okd} idx={oki}{flag}", flush=True)
print("mismatches:", bad, "time", round(time.time() - t_all, 1))
sys.exit(1 if bad else 0)
