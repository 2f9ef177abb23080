"""Seeded synthetic inputs shared by bench.py, the tests and the golden-vector
generator: GBM log-returns (i.i.d. Gaussian), the dataset the benchmark
configurations in BASELINE.json are quoted on."""
from __future__ import annotations

import hashlib

import numpy as np

MU, SIGMA, DT = 0.05, 0.2, 1.0 / 252.0
DATASET_SEED, QUERY_SEED = 0, 1


def gbm_log_returns(shape, seed: int) -> np.ndarray:
    """float32 log-returns (mu - sigma^2/2) dt + sigma sqrt(dt) Z, Z ~ N(0,1)
    drawn from numpy's default_rng(seed) in C order."""
    z = np.random.default_rng(seed).standard_normal(shape)
    return ((MU - SIGMA ** 2 / 2) * DT + SIGMA * np.sqrt(DT) * z).astype(np.float32)


def dataset(R: int, T: int, seed: int = DATASET_SEED) -> np.ndarray:
    """(R, 1, T) trajectory ensemble."""
    return gbm_log_returns((R, 1, T), seed)


def dataset_rows(R: int, T: int, seed: int, row_start: int, row_stop: int) -> np.ndarray:
    """Rows [row_start, row_stop) of dataset(R, T, seed) without materialising
    the rest (the generator stream is consumed row by row, so a rank can build
    its own shard)."""
    rng = np.random.default_rng(seed)
    out = np.empty((row_stop - row_start, 1, T), np.float32)
    chunk = 1024
    r = 0
    while r < row_stop:
        n = min(chunk, row_stop - r)
        z = rng.standard_normal((n, 1, T))
        lo, hi = max(r, row_start), min(r + n, row_stop)
        if lo < hi:
            out[lo - row_start:hi - row_start] = (
                (MU - SIGMA ** 2 / 2) * DT + SIGMA * np.sqrt(DT) * z[lo - r:hi - r]
            ).astype(np.float32)
        r += n
    return out


def single_query(W: int, seed: int = QUERY_SEED) -> np.ndarray:
    """(W,) an independent draw: an "unseen history"."""
    return gbm_log_returns((W,), seed)


def rolling_queries(B: int, W: int, seed: int = QUERY_SEED) -> np.ndarray:
    """(B, W) stride-1 rolling windows over one extra path of length W+B-1
    (the batched "query dates" configuration)."""
    path = gbm_log_returns((W + B - 1,), seed)
    return np.lib.stride_tricks.sliding_window_view(path, W).copy()


def wavelet_bank(n_scales: int, K: int, xi: float = 2.35) -> np.ndarray:
    """(2 * n_scales + 1, K) float32 real filter bank for BASELINE.json configs[4] ("wavelet
    conv, W=252"): cosine and sine parts of Morlet wavelets at dyadic scales 2^1 .. 2^n_scales
    plus one Gaussian low-pass, each anchored at the END of the window (the most recent
    samples) and L1-normalised -- the linear stage of a scattering embedding.  (Its modulus /
    spectra stages live in the reference's un-vendored dependency and are out of reach.)"""
    u = np.arange(K, dtype=np.float64)[::-1]           # 0 = the newest sample
    rows = []
    for j in range(1, n_scales + 1):
        sigma = 0.8 * 2.0 ** j
        env = np.exp(-0.5 * ((u - 3 * sigma) / sigma) ** 2)
        for phase in (np.cos, np.sin):
            w = env * phase(xi * (u - 3 * sigma) / 2.0 ** j)
            w -= env * (w.sum() / env.sum())            # zero mean
            rows.append(w / np.abs(w).sum())
    sigma = 0.8 * 2.0 ** n_scales
    low = np.exp(-0.5 * ((u - 3 * sigma) / sigma) ** 2)
    rows.append(low / low.sum())
    return np.asarray(rows, dtype=np.float32)


def sha256(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
