"""Shape fuzz of psh_scan_topk (Identity + RelativeMSE) through the status protocol (scan_topk_checked): every window length
1 .. 256, 1 .. 20 queries, ragged / unaligned rows, horizons, k from 1 to thousands, the adversarial kinds of tests/_adversarial.py
and SMOOTH ensembles (random walks: clustered matches, block lists that fill up), with and without PSH_FLAG_OVERLAP and with
admission hints (good, and one query's far too low) -- whatever launch structure the library picks (fused launch, the three
launches with the short or the long-window scan, the batched kernels, the loops of steps for batches with long windows, the
exhaustive path for small problems), HIP vs the CPU oracle, bit for bit.     python tests/stress/stress_shapes.py SEED CASES"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import oracle
from _adversarial import KINDS, make
from shadowing_amd import _native

def run(seed: int, n_cases: int, verbose: bool = True):
    """(mismatches, {path: cases}) of `n_cases` random cases."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    bad = 0
    paths_seen = {}
    t_all = time.time()
    for case in range(n_cases):
        R = int(rng.choice([1500, 2048, 3001, 4096, 8192])); T = int(rng.choice([700, 1024, 1500, 2048, 2051, 3000]))
        W = int(rng.choice([1, 5, 8, 16, 17, 20, 24, 25, 26, 30, 33, 34, 40, 64, 100, 126, 200, 252, 256]))
        W = min(W, T // 3)
        h = int(rng.choice([0, 0, 1, 7, 20, 60]))
        B = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 7, 16, 20]))
        k = int(rng.choice([1, 10, 64, 200, 1024, 3000]))
        Tp = T - W - h + 1
        if Tp < 1:
            continue
        k = min(k, R * Tp)
        cseed = int(rng.integers(1 << 30))
        kind = str(rng.choice(list(KINDS) + ["walk", "walk"]))
        if kind == "walk":
            g = np.random.default_rng(cseed)
            ds = (0.05 * np.cumsum(g.standard_normal((R, T)), axis=1)).astype(np.float32)
            q = (0.05 * np.cumsum(g.standard_normal((B, W)), axis=1)).astype(np.float32)
        else:
            ds, q = make(kind, R, T, B, W, h, cseed)
        if rng.random() < 0.3:                                  # an unaligned view: rows that do not start on 16 bytes
            big = np.zeros((R, T + 3), np.float32); big[:, 1:T + 1] = ds
            ds_t = torch.as_tensor(big).to(dev)[:, 1:T + 1]
            ds_t = ds_t.contiguous() if rng.random() < 0.5 else torch.as_tensor(ds).to(dev)
        else:
            ds_t = torch.as_tensor(ds).to(dev)
        q_t = torch.as_tensor(q).to(dev)
        od, oidx = oracle.scan_topk(ds[:, None, :], q, k, h=h)
        flags = int(rng.choice([0, 0, _native.FLAG_OVERLAP]))
        info = {}
        _native.scan_topk(ds_t, q_t, k, h=h, flags=flags, info=info)
        torch.cuda.synchronize()
        paths_seen[info["path"]] = paths_seen.get(info["path"], 0) + 1
        modes = ["plain"]
        if np.isfinite(od[:, k - 1]).all() and (od[:, k - 1] > 0).all():
            modes += ["hint", "short_hint"]
        for mode in modes:
            hint = None
            if mode != "plain":
                lev = ((od[:, k - 1].astype(np.float64) ** 2) * (q.astype(np.float64) ** 2).sum(axis=1) * 1.2).astype(np.float32)
                if mode == "short_hint":
                    lev[int(rng.integers(0, B))] *= 1e-4
                if not (np.isfinite(lev).all() and (lev > 0).all()):
                    continue
                hint = torch.as_tensor(lev).to(dev)
            d, idx = _native.scan_topk_checked(ds_t, q_t, k, h=h, flags=flags, tau_hint=hint)
            d, idx = d.cpu().numpy(), idx.cpu().numpy()
            same_d = np.array_equal(d.view(np.uint32), od.view(np.uint32)) or (np.isnan(d) == np.isnan(od)).all() and np.array_equal(d[~np.isnan(d)].view(np.uint32), od[~np.isnan(od)].view(np.uint32))
            # (NaN distances tie: their indices may come in any order)
            fin = ~np.isnan(od)
            same_i = np.array_equal(idx[fin], oidx[fin])
            if not (same_d and same_i):
                bad += 1
                print(f"MISMATCH case {case} {mode}: kind={kind} R={R} T={T} W={W} h={h} B={B} k={k} flags={flags} path={info['path']} seed={cseed}", flush=True)
        if verbose and case % 10 == 9:
            print(f"... {case + 1} cases, {bad} mismatches, paths {paths_seen}, {time.time() - t_all:.0f} s", flush=True)
    return bad, paths_seen


if __name__ == "__main__":
    t0 = time.time()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    nbad, seen = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, n)
    print(f"stress_shapes: {n} cases, {nbad} mismatches, paths taken by the plain call {seen}, {time.time() - t0:.0f} s")
    sys.exit(1 if nbad else 0)
