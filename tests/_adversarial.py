"""Adversarial ensembles / query batches for the rejection filters (shared by tests/test_gpu_admitted_set.py,
tests/test_gpu_batched.py's promoted stress cases and tests/stress/stress_mq8.py): what a rigorous lower bound of acc has to
survive -- planted near-matches and exact copies of the queries (true neighbours with tiny distances: none of them may be
rejected), spikes no sample ever saw, queries far smaller / larger than the data, amplitudes spread inside the one
quantisation step of a batch, extreme fp32 scales (f16 overflow on one side, subnormal squares on the other), heavy tails,
zero / constant / loud rows, quiet stretches."""
from __future__ import annotations

import numpy as np

from shadowing_amd import synthetic as syn

KINDS = ("spikes", "tiny_queries", "huge_queries", "scale_up", "scale_down", "planted_matches", "student_t", "zero_constant_rows",
         "loud_rows", "quiet_one_loud", "spread_amplitudes", "quiet_stretches", "plain")


def make(kind: str, R: int, T: int, B: int, W: int, h: int, seed: int):
    """(ds (R, T) float32, q (B, W) float32) of one adversarial kind, seeded."""
    assert kind in KINDS, kind
    rng = np.random.default_rng(seed)
    ds = syn.dataset(R, T, seed)[:, 0, :].copy()
    q = syn.rolling_queries(B, W, seed + 1) if (seed & 1) else syn.gbm_log_returns((B, W), seed + 1)
    q = np.ascontiguousarray(q, dtype=np.float32).reshape(B, W)
    if kind == "spikes":                       # isolated values 1e2 .. 1e6 x the rest (f16 overflow after scaling)
        for r in rng.integers(0, R, 6):
            ds[r, rng.integers(0, T, 5)] *= float(10.0 ** rng.integers(2, 7))
    elif kind == "tiny_queries":
        q *= float(10.0 ** -rng.integers(2, 6))
    elif kind == "huge_queries":
        q *= float(10.0 ** rng.integers(2, 5))
    elif kind == "scale_up":                   # everything near the top of fp32's useful range for squares
        s = float(10.0 ** rng.integers(8, 15)); ds *= s; q *= s
    elif kind == "scale_down":                 # ... and near the bottom (squares go subnormal)
        s = float(10.0 ** -rng.integers(8, 15)); ds *= s; q *= s
    elif kind == "planted_matches":            # near-matches (relative noise 1e-1 .. 1e-5) and exact copies of the queries
        for b in rng.integers(0, B, 60):
            r = int(rng.integers(0, R)); t = int(rng.integers(0, T - W - h))
            ds[r, t:t + W] = q[b] * (1 + float(10.0 ** -rng.integers(1, 6)) * rng.standard_normal(W).astype(np.float32))
        for b in rng.integers(0, B, 10):
            ds[int(rng.integers(0, R)), 5:5 + W] = q[b]
    elif kind == "student_t":
        ds = (0.01 * rng.standard_t(2.5, size=ds.shape)).astype(np.float32)
    elif kind == "zero_constant_rows":
        ds[::7] = 0; ds[3::11] = 0.01; q[:, ::3] = 0
    elif kind == "loud_rows":
        ds[R // 3: R // 3 + 40] *= 1000.0
    elif kind == "quiet_one_loud":
        ds *= 1e-3; ds[1] *= 1e5
    elif kind == "spread_amplitudes":          # up to 30x inside the batch's ONE quantisation step
        q *= rng.uniform(1.0, 30.0, (B, 1)).astype(np.float32)
    elif kind == "quiet_stretches":            # a third of every row far below the batch's scale; constant stretches
        ds[:, : T // 3] *= 1e-5; ds[::5, T // 2: T // 2 + 300] = ds[::5, T // 2: T // 2 + 1]
    return np.ascontiguousarray(ds, dtype=np.float32), q
