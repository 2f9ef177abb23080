"""Adversarial parity stress of the batched scan's 8-BIT rejection test (scan_mq8_kernel: 32 queries and more, W <= 25, sampled
path): what its quantisation bound has to survive -- planted near-matches and exact copies of the queries (true neighbours with
tiny distances: nothing of them may be rejected), spikes the bootstrap never saw, queries far smaller / larger than the data,
amplitudes spread inside the one step of a batch, extreme fp32 scales, heavy tails, zero / constant / loud rows, quiet
segments, run-time window lengths.  HIP scan (default: the 8-bit test) vs the CPU oracle, bit for bit; every third case also the
f16 test (PSH_FLAG_MQ_F16) on the same inputs.     python tests/stress/stress_mq8.py SEED CASES"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import oracle
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
t_all = time.time()


def run(ds_t, q_t, k, h, flags):
    d, idx, st, prof = _native.scan_topk(ds_t, q_t, k, h=h, profile=True, flags=flags)
    torch.cuda.synchronize()
    badq = torch.nonzero(st != 0).flatten()
    if badq.numel():
        d2, i2, _ = _native.scan_topk(ds_t, q_t[badq].contiguous(), k, h=h, exhaustive=True)
        d[badq] = d2; idx[badq] = i2
    return d.cpu().numpy(), idx.cpu().numpy(), int(badq.numel()), prof


for case in range(n_cases):
    R = int(rng.choice([2048, 3000, 4096, 6000])); T = int(rng.choice([768, 1024, 1500, 2048]))
    W = int(rng.choice([20, 20, 20, 8, 13, 17, 25])); h = int(rng.integers(0, 30)); k = int(rng.choice([1, 32, 200, 1024]))
    B = int(rng.choice([32, 33, 48, 64, 100, 130, 257]))
    seed = int(rng.integers(1 << 30))
    ds = syn.dataset(R, T, seed)[:, 0, :].copy()
    q = syn.rolling_queries(B, W, seed + 1) if rng.random() < 0.5 else syn.gbm_log_returns((B, W), seed + 1)
    q = np.ascontiguousarray(q, dtype=np.float32)
    kind = case % 12
    if kind == 0:
        for r in rng.integers(0, R, 6): ds[r, rng.integers(0, T, 5)] *= float(10.0 ** rng.integers(2, 7))
        note = "spikes"
    elif kind == 1:
        q *= float(10.0 ** -rng.integers(2, 6)); note = "tiny queries"
    elif kind == 2:
        q *= float(10.0 ** rng.integers(2, 5)); note = "huge queries"
    elif kind == 3:
        s = float(10.0 ** rng.integers(-15, 15)); ds *= s; q *= s; note = f"scale {s:g}"
    elif kind == 4:   # near-matches and exact copies of the queries inside the data
        for b in rng.integers(0, B, 60):
            r = int(rng.integers(0, R)); t = int(rng.integers(0, T - W - h))
            ds[r, t:t + W] = q[b] * (1 + float(10.0 ** -rng.integers(1, 6)) * rng.standard_normal(W).astype(np.float32))
        for b in rng.integers(0, B, 10):
            ds[int(rng.integers(0, R)), 5:5 + W] = q[b]
        note = "planted matches"
    elif kind == 5:
        ds = (0.01 * rng.standard_t(2.5, size=ds.shape)).astype(np.float32); note = "student-t(2.5)"
    elif kind == 6:
        ds[::7] = 0; ds[3::11] = 0.01; q[:, ::3] = 0; note = "zero/constant rows"
    elif kind == 7:
        ds[R // 3: R // 3 + 40] *= 1000.0; note = "loud rows"
    elif kind == 8:
        ds *= 1e-3; ds[1] *= 1e5; note = "quiet + one loud row"
    elif kind == 9:   # amplitudes spread inside the batch's one step (up to 3x: what PathShadowing sends in one call; then 30x)
        q *= rng.uniform(1.0, 3.0 if case % 24 == 9 else 30.0, (B, 1)).astype(np.float32); note = "spread amplitudes"
    elif kind == 10:  # quiet stretches inside the rows (a segment far below the batch's scale), constant stretches
        ds[:, : T // 3] *= 1e-5; ds[::5, T // 2: T // 2 + 300] = ds[::5, T // 2: T // 2 + 1]; note = "quiet / constant stretches"
    else:
        note = "plain"
    ds_t = torch.as_tensor(ds).to(dev); q_t = torch.as_tensor(q).to(dev)
    d, idx, novf, prof = run(ds_t, q_t, k, h, 0)
    od, oidx = oracle.scan_topk(ds, q, k, h=h)
    ok = np.array_equal(d.view(np.uint32), od.view(np.uint32)) and np.array_equal(idx, oidx)
    ok16 = True
    if case % 3 == 0:
        d16, i16, _, _ = run(ds_t, q_t, k, h, _native.FLAG_MQ_F16)
        ok16 = np.array_equal(d16.view(np.uint32), od.view(np.uint32)) and np.array_equal(i16, oidx)
    bad += not (ok and ok16)
    print(f"{note:26s} R={R} T={T} W={W} h={h} k={k} B={B} path={prof['path']} cand={prof['n_candidates']} rerun={novf} "
          f"8-bit={ok} f16={ok16}{'' if ok and ok16 else '   <<<<<< MISMATCH'}", flush=True)
print("mismatches:", bad, "time", round(time.time() - t_all, 1))
sys.exit(1 if bad else 0)
