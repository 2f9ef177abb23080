"""Adversarial A/B of the dense embedded scan with its rejection test on the matrix cores (PSH_FLAG_EMBED_MX) against the same
scan on the vector ALUs (the exact dense chains for every window), bit for bit: random dense kernels (d <= 12, K <= 256), batches
whose queries differ by many orders of magnitude (ONE f16 scale serves the batch), coordinates far below a query's largest,
extreme common scales, spikes and NaNs in the data, planted near-matches.  Every 5th case also against the CPU oracle."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import oracle
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t_all = time.time()


def run(dsd, kd, hd, k, h, flags):
    d, idx, st = _native.scan_topk_embedded(dsd, kd, hd, k, h=h, flags=flags)
    badq = torch.nonzero(st != 0).flatten()
    if badq.numel():
        d2, i2, _ = _native.scan_topk_embedded(dsd, kd, hd[badq].contiguous(), k, h=h, flags=flags, exhaustive=True)
        d[badq] = d2; idx[badq] = i2
    return d.cpu().numpy(), idx.cpu().numpy(), int(badq.numel())


for case in range(n_cases):
    K = int(rng.choice([8, 20, 33, 64, 100, 126, 200, 252, 256])); dim = int(rng.integers(1, 13))
    ker = (rng.standard_normal((dim, K)) * rng.uniform(0.01, 1.0, (dim, 1))).astype(np.float32)
    if rng.random() < 0.5:                                                  # band-limited rows (wavelet-like supports)
        for i in range(dim):
            w = int(rng.integers(1, K + 1)); ker[i, : K - w] = 0
    if rng.random() < 0.3: ker[rng.integers(0, dim)] *= np.float32(10.0 ** rng.integers(-6, 4))
    T = int(rng.integers(K + 60, 2600)); h = int(rng.integers(0, 25))
    R = int(rng.choice([400, 1024, 2500])); B = int(rng.choice([1, 2, 3, 5, 16, 37]))
    Tp = T - K - h + 1
    k = int(min(rng.choice([1, 20, 300, 1500]), R * Tp // 4))
    seed = int(rng.integers(1 << 30))
    ds = syn.dataset(R, T, seed)[:, 0, :].copy()
    x = syn.gbm_log_returns((B, K), seed + 1)
    kind = case % 8
    note = "plain"
    if kind == 0:
        sc = (10.0 ** rng.integers(-6, 5, size=B)).astype(np.float32); x *= sc[:, None]; note = "queries over 11 decades"
    elif kind == 1:
        x[0] *= 1e-7; x[-1] *= 3e3; note = "one tiny, one loud query"
    elif kind == 2:
        s = float(10.0 ** rng.integers(-12, 12)); ds *= s; x *= s; note = f"scale {s:g}"
    elif kind == 3:
        for r in rng.integers(0, R, 5): ds[r, rng.integers(0, T, 4)] *= float(10.0 ** rng.integers(2, 6))
        ds[rng.integers(0, R), rng.integers(0, T)] = np.nan; note = "spikes + NaN"
    elif kind == 4:
        for r in rng.integers(0, R, 40):
            t = int(rng.integers(0, Tp)); b = int(rng.integers(0, B))
            ds[r, t:t + K] = x[b] * (1 + 1e-3 * rng.standard_normal(K).astype(np.float32))
        ds[3, 10:10 + K] = x[0]; note = "planted matches"
    elif kind == 5:
        ds = (0.01 * rng.standard_t(2.5, size=ds.shape)).astype(np.float32); note = "student-t(2.5)"
    elif kind == 6:
        ds[::5] = 0; x[:, ::2] = 0; note = "zero rows, zeros in the queries"
    kt, xt = torch.tensor(ker), torch.tensor(x)
    hx = torch.nn.functional.conv1d(xt[:, None, :], kt[:, None, :])[:, :, 0].contiguous()
    dsd, kd, hd = torch.tensor(ds).to(dev), kt.to(dev), hx.to(dev)
    d1, i1, o1 = run(dsd, kd, hd, k, h, _native.FLAG_EMBED_MX)
    d0, i0, o0 = run(dsd, kd, hd, k, h, _native.FLAG_EMBED_DENSE)
    ok = np.array_equal(d1.view(np.uint32), d0.view(np.uint32)) and np.array_equal(i1, i0)
    if ok and case % 5 == 0 and R * Tp * B * dim * K < 3e11:
        od, oi = oracle.scan_topk_embedded(ds[:, None, :], ker, hx.numpy(), k, h=h)
        ok = np.array_equal(d1.view(np.uint32), od.view(np.uint32)) and np.array_equal(i1, oi)
        note += " (+oracle)"
    bad += not ok
    print(f"{note:34s} d={dim} K={K} R={R} T={T} h={h} B={B} k={k} exhaustive mx/dense={o1}/{o0} {'ok' if ok else '<<<<<< MISMATCH'}", flush=True)
print("mismatches:", bad, "time", round(time.time() - t_all, 1))
sys.exit(1 if bad else 0)
