"""Parity tests proper: the HIP path, called through the C ABI (ctypes binding
shadowing_amd._native), against the CPU oracle on the same seeded inputs, against the
committed golden vectors of the reference, and -- at BASELINE.json's full sizes --
through size-independent properties.  Bar: distances BIT-exact, indices identical."""
import numpy as np
import pytest
import torch

from _util import (BIG_GOLDENS, SMALL_GOLDENS, assert_exact, assert_matches_reference, bits, canonical, load_golden,
                   rows3)
from shadowing_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def hip_scan(dev, ds, q, k, h, **kw):
    from shadowing_amd import _native
    ds_t = torch.as_tensor(np.ascontiguousarray(rows3(ds)[:, 0, :])).to(dev)
    q_t = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(q), dtype=np.float32)).to(dev)
    out = _native.scan_topk(ds_t, q_t, k, h=h, **kw)
    torch.cuda.synchronize(dev)
    return out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy(), out[3:] and out[3]


@pytest.mark.parametrize("name", SMALL_GOLDENS)
def test_hip_matches_reference_goldens_small(hip_device, oracle_mod, name):
    g = load_golden(name)
    ds = rows3(g["dataset"])
    h = g["h"] or 0
    d, idx, status, _ = hip_scan(hip_device, ds, g["queries"], g["k"], h)
    assert np.all(status == 0)
    all_dist = [oracle_mod.all_distances(ds, q, h) for q in g["queries"]]
    assert_matches_reference(d, idx, g, all_dist, what=name)
    od, oidx = oracle_mod.scan_topk(ds, g["queries"], g["k"], h=h)
    assert_exact(d, idx, od, oidx, name + " vs oracle")


@pytest.mark.parametrize("name", BIG_GOLDENS)
def test_hip_matches_reference_goldens_full_size(hip_device, name):
    """BASELINE.json configs[1] (R=32768, T=4096, W=20, k=1024) and the rolling-query
    shape of configs[2]: the reference's own CPU output, regenerated dataset checked by
    SHA-256."""
    g = load_golden(name)
    d, idx, status, _ = hip_scan(hip_device, g["dataset"], g["queries"], g["k"], g["h"] or 0)
    assert np.all(status == 0)
    assert_matches_reference(d, idx, g, None, what=name)


CASES = [  # R, T, W, h, k, B
    (64, 1024, 20, 20, 64, 1),
    (300, 1100, 20, 20, 128, 3),      # ragged last segment (T' = 1061)
    (17, 4096, 20, 0, 1000, 2),
    (50, 515, 20, 7, 33, 2),          # T % 4 != 0: unaligned rows
    (40, 600, 8, 3, 50, 2),           # runtime-W kernel, W < 16
    (40, 600, 16, 0, 50, 2),          # W == 16: exactly one block
    (40, 600, 37, 5, 50, 2),          # W = 2 blocks + 5
    (24, 900, 256, 10, 20, 1),        # PSH_MAX_W
    (6, 2100, 20, 20, 1, 1),          # k = 1
    (2048, 512, 20, 20, 777, 4),
    (4096, 1024, 20, 20, 5000, 2),    # kpad = 8192: merge sort by ranking, both buffers in LDS
    (4096, 1024, 20, 20, 10000, 2),   # kpad = 16384 (the reference test's k): second buffer in global scratch
    (2048, 2048, 20, 0, 16384, 1),    # PSH_MAX_K
]


@pytest.mark.parametrize("R,T,W,h,k,B", CASES)
def test_hip_equals_oracle_seeded(hip_device, oracle_mod, R, T, W, h, k, B):
    ds = syn.dataset(R, T, 100 + R)
    q = syn.gbm_log_returns((B, W), 200 + W)
    d, idx, status, _ = hip_scan(hip_device, ds, q, k, h)
    assert np.all(status == 0)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, f"R={R} T={T} W={W} h={h} k={k} B={B}")


@pytest.mark.parametrize("R,T,W,h,k,B", [(2048, 512, 20, 20, 777, 4), (300, 1100, 20, 20, 128, 3), (40, 600, 37, 5, 50, 2)])
def test_exhaustive_path_equals_oracle(hip_device, oracle_mod, R, T, W, h, k, B):
    ds = syn.dataset(R, T, 100 + R)
    q = syn.gbm_log_returns((B, W), 200 + W)
    d, idx, status, _ = hip_scan(hip_device, ds, q, k, h, exhaustive=True)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, "exhaustive")


def test_sampled_path_is_taken_and_profiled(hip_device, oracle_mod):
    """Big enough that the threshold path (sample -> tau -> filter -> select) runs."""
    R, T, W, h, k = 4096, 4096, 20, 20, 1024
    ds = syn.dataset(R, T, 7)
    q = syn.single_query(W, 8)
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, h, profile=True)
    assert prof["path"] == 0 and prof["n_sample_rows"] > 0 and prof["scan_ms"] > 0
    assert np.all(status == 0)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, "sampled path")


def test_massive_ties_overflow_is_reported_and_exhaustive_is_exact(hip_device, oracle_mod):
    """A constant dataset: every window ties.  The threshold path must flag the
    overflow (never return silently wrong rows); the exhaustive path returns the
    canonical first-k windows, as the oracle does."""
    R, T, W, h, k = 4096, 1024, 20, 20, 64
    ds = np.full((R, 1, T), 0.01, np.float32)
    q = syn.single_query(W, 9)
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, h, profile=True)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    if prof["path"] == 0:
        assert np.all(status == 1), "overflow must be reported"
    d2, idx2, _, _ = hip_scan(hip_device, ds, q, k, h, exhaustive=True)
    assert_exact(d2, idx2, od, oidx, "ties, exhaustive")
    # and the host wrapper resolves it transparently
    import shadowing_amd as sa
    obj = sa.PathShadowing(sa.Identity(W), sa.RelativeMSE(), ds, sa.PredictionContext(h), cache=True)
    d3, _, idx3 = obj.shadow(q, k=k, cuda=True)
    assert_exact(d3, idx3, od, oidx, "ties, PathShadowing")


def test_query_norm_kernel(hip_device, oracle_mod):
    from shadowing_amd import _native
    rng = np.random.default_rng(3)
    for W in list(range(1, 40)) + [63, 64, 126, 256]:
        x = (rng.standard_normal((5, W)) * 0.0126).astype(np.float32)
        got = _native.query_norm(torch.as_tensor(x).to(hip_device)).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle_mod.qnorm(x))), W


def test_shard_and_merge_invariance(hip_device, oracle_mod):
    """Scanning row shards with r_offset and merging on the device equals the whole."""
    from shadowing_amd import _native
    R, T, W, h, k, B = 1000, 800, 20, 20, 300, 3
    ds = syn.dataset(R, T, 11)
    q = syn.gbm_log_returns((B, W), 12)
    whole = hip_scan(hip_device, ds, q, k, h)
    parts_d, parts_i = [], []
    for lo, hi in ((0, 1), (1, 400), (400, 1000)):    # first shard has fewer than k windows? (1 row: 741) no -> use k_local
        n_local = (hi - lo) * (T - W - h + 1)
        kl = min(k, n_local)
        ds_t = torch.as_tensor(ds[lo:hi, 0, :].copy()).to(hip_device)
        dd, ii, st = _native.scan_topk(ds_t, torch.as_tensor(q).to(hip_device), kl, h=h, r_offset=lo)
        if kl < k:
            dd = torch.cat([dd, dd.new_full((B, k - kl), float("inf"))], 1)
            ii = torch.cat([ii, ii.new_full((B, k - kl, 2), -1)], 1)
        parts_d.append(dd)
        parts_i.append(ii)
    md, mi = _native.merge_topk(torch.cat(parts_d, 1).contiguous(), torch.cat(parts_i, 1).contiguous(), k)
    torch.cuda.synchronize()
    assert_exact(md.cpu().numpy(), mi.cpu().numpy(), whole[0], whole[1], "sharded + merged")
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(whole[0], whole[1], od, oidx, "whole")


def test_gather_paths_kernel(hip_device, oracle_mod):
    from shadowing_amd import _native
    ds = syn.dataset(50, 300, 13)
    q = syn.gbm_log_returns((2, 20), 14)
    d, idx = oracle_mod.scan_topk(ds, q, 25, h=20)
    got = _native.gather_paths(torch.as_tensor(ds).to(hip_device), torch.as_tensor(idx).to(hip_device), 40)
    assert np.array_equal(got.cpu().numpy()[:, :, 0, :], oracle_mod.gather_paths(ds, idx, 40))


def test_scale_invariance_of_indices(hip_device):
    """RelativeMSE is scale-free: dataset*c, query*c (c a power of two: exact) gives the
    same indices and bit-identical distances."""
    ds = syn.dataset(512, 1024, 15)
    q = syn.single_query(20, 16)
    a = hip_scan(hip_device, ds, q, 200, 20)
    b = hip_scan(hip_device, ds * 4.0, q * 4.0, 200, 20)
    assert_exact(b[0], b[1], a[0], a[1], "scale invariance")


def test_full_size_properties(hip_device):
    """configs[1] size: rows sorted, indices admissible and distinct, every returned
    distance re-derivable from the dataset, halves + merge == whole."""
    from shadowing_amd import _native
    g = load_golden("cfg2_R32768")
    ds, q, k, h, W = g["dataset"], g["queries"], g["k"], g["h"], g["W"]
    d, idx, status, _ = hip_scan(hip_device, ds, q, k, h)
    assert np.all(status == 0) and np.all(np.diff(d[0]) >= 0)
    assert idx[..., 0].min() >= 0 and idx[..., 0].max() < ds.shape[0]
    assert idx[..., 1].min() >= 0 and idx[..., 1].max() <= ds.shape[-1] - W - h
    assert len({tuple(v) for v in idx[0]}) == k
    # recompute each returned distance on the host in the reference's order
    xn = np.float32(g["xn"][0])
    for j in range(0, k, 37):
        r, t = idx[0, j]
        acc = np.float32(0)
        for i in range(W):
            D = np.float32(q[0, i] - ds[r, 0, t + i])
            acc = np.float32(np.float64(D) * np.float64(D) + np.float64(acc))
        assert bits(np.float32(np.sqrt(acc)) / xn) == bits(d[0, j])


def test_merge_of_gathered_lists_in_place(hip_device, oracle_mod):
    """What the multi-GPU path does after its one all-gather, emulated on one GPU: three
    row shards scanned with r_offset, their (d | idx) send buffers laid out rank-major as
    all_gather_into_tensor would leave them, merged in place."""
    from shadowing_amd import _native
    R, T, W, h, k, B = 900, 700, 20, 20, 256, 2
    ds = syn.dataset(R, T, 21)
    q = syn.gbm_log_returns((B, W), 22)
    shards = ((0, 300), (300, 650), (650, 900))
    gathered = torch.empty((len(shards), 3 * B * k), dtype=torch.int32, device=hip_device)
    for g, (lo, hi) in enumerate(shards):
        send = gathered[g]
        out = (send[:B * k].view(torch.float32).view(B, k), send[B * k:].view(B, k, 2))
        _native.scan_topk(torch.as_tensor(ds[lo:hi, 0, :].copy()).to(hip_device), torch.as_tensor(q).to(hip_device),
                          k, h=h, r_offset=lo, out=out)
    md, mi = _native.merge_topk_gathered(gathered, len(shards), B, k, k)
    torch.cuda.synchronize()
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(md.cpu().numpy(), mi.cpu().numpy(), od, oidx, "gathered merge")


@pytest.mark.parametrize("exchange", ["library", "torch"])
def test_sharded_class_with_rccl_process_group(hip_device, oracle_mod, tmp_path, exchange):
    """ShardedPathShadowing end to end over RCCL (one rank: the all-gather and the merge still run) -- with the
    collective issued by libpsh_hip.so itself (psh_exchange_merge: side stream, one event hand-over) and by
    torch.distributed; then three query batches pipelined (batch i+1 scanned before batch i is finished)."""
    import torch.distributed as dist
    import shadowing_amd as sa
    from shadowing_amd.distributed import ShardedPathShadowing
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1, device_id=hip_device)
    obj = None
    try:
        R, T, W, h, k, B = 600, 900, 20, 20, 128, 3
        ds = syn.dataset(R, T, 23)
        q = syn.rolling_queries(B, W, 24)
        obj = ShardedPathShadowing(sa.Identity(W), sa.RelativeMSE(), ds, 0, sa.PredictionContext(h),
                                   device=hip_device, always_exchange=True, exchange=exchange)
        d, paths, idx = obj.shadow(q, k)
        od, opaths, oidx = oracle_mod.shadow(ds, q, k, h)
        assert_exact(d, idx, od, oidx, "sharded class")
        assert np.array_equal(paths, opaths)
        assert (obj._comm is not None) == (exchange == "library")
        # a larger shard (the fused single-query launch with reserved CUs), pipelined
        big = syn.dataset(8192, 2048, 25)
        obj2 = ShardedPathShadowing(sa.Identity(W), sa.RelativeMSE(), big, 0, sa.PredictionContext(h),
                                    device=hip_device, always_exchange=True, exchange=exchange)
        qs = [torch.tensor(syn.gbm_log_returns((1, W), 26 + i)) for i in range(4)]
        outs, pend = [], None
        for qi in qs:
            nxt = obj2.scan_begin(qi, 256)
            if pend is not None:
                outs.append(pend.finish())
            pend = nxt
        outs.append(pend.finish())
        torch.cuda.synchronize()
        for qi, (dd, ii) in zip(qs, outs):
            od2, oi2 = oracle_mod.scan_topk(big, qi.numpy(), 256, h=h)
            assert_exact(dd.cpu().numpy(), ii.cpu().numpy(), od2, oi2, "pipelined batch")
        obj2.close()
    finally:
        if obj is not None:
            obj.close()
        dist.destroy_process_group()


# ---- the matrix-core rejection test (scan_mx_kernel: W = 20, one query, sampled path) ----------------------
def _adversarial(kind, R, T, seed):
    rng = np.random.default_rng(seed)
    ds = syn.dataset(R, T, seed)[:, 0, :].copy()
    q = syn.gbm_log_returns((1, 20), seed + 1)
    if kind == "spikes":                      # outliers in rows the bootstrap sample mostly never sees
        for r in rng.integers(0, R, 8):
            ds[r, rng.integers(0, T, 5)] *= float(10.0 ** rng.integers(2, 7))
        ds[8, 100] *= 1e5; ds[24, 1500] *= 3e3       # rows the bootstrap does visit (its f16-overflow fallback)
    elif kind == "tiny_query":
        q *= 1e-4
    elif kind == "huge_query":
        q *= 1e3
    elif kind == "scale_1e-12":
        ds *= 1e-12; q *= 1e-12
    elif kind == "scale_1e+12":
        ds *= 1e12; q *= 1e12
    elif kind == "planted":                   # near-copies and one exact copy of the query
        for r in rng.integers(0, R, 40):
            t = int(rng.integers(0, T - 20))
            ds[r, t:t + 20] = q[0] * (1 + 1e-3 * rng.standard_normal(20).astype(np.float32))
        ds[7, 100:120] = q[0]
    elif kind == "student_t":
        ds = (0.01 * rng.standard_t(2.5, size=ds.shape)).astype(np.float32)
    elif kind == "zero_rows":
        ds[::7] = 0; ds[3::11] = 0.01; q[0, ::3] = 0
    elif kind == "one_loud_row":              # everything quiet, one unsampled row 1e5 x louder: f16 overflow
        ds *= 1e-3; ds[1] *= 1e5
    elif kind == "f16_overflow_inf":          # values whose scaled square leaves fp32/f16 range altogether
        ds[5, 300:310] = 1e30; ds[9, 50] = -3e38
    return ds, q


MX_KINDS = ["spikes", "tiny_query", "huge_query", "scale_1e-12", "scale_1e+12", "planted", "student_t", "zero_rows",
            "one_loud_row", "f16_overflow_inf"]


@pytest.mark.parametrize("kind", MX_KINDS)
def test_matrix_core_filter_is_exact_on_adversarial_data(hip_device, oracle_mod, kind):
    """The f16 rejection test may only ever skip windows that provably cannot be admitted:
    results stay bit-exact whatever the magnitudes."""
    R, T, h, k = 6000, 2048, 11, 700
    ds, q = _adversarial(kind, R, T, 900 + MX_KINDS.index(kind))
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, h, profile=True)
    assert prof["path"] == 0                                    # the sampled path (the one that filters)
    if status[0] != 0:                                          # legitimately overflowed (massive ties): exact net
        d, idx, _, _ = hip_scan(hip_device, ds, q, k, h, exhaustive=True)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, kind)


def test_matrix_core_and_valu_filters_admit_the_same_windows(hip_device):
    ds = syn.dataset(8192, 4096, 77)
    q = syn.single_query(20, 78)
    from shadowing_amd import _native
    d1, i1, s1, p1 = hip_scan(hip_device, ds, q, 1024, 20, profile=True)
    d2, i2, s2, p2 = hip_scan(hip_device, ds, q, 1024, 20, profile=True, flags=_native.FLAG_FILTER_VALU)
    assert s1[0] == 0 and s2[0] == 0
    assert_exact(d1, i1, d2, i2, "mx vs valu filter")
    # the matrix-core scan files its candidates in two classes and reports the first (below an ESTIMATE of the k-th
    # smallest, x2); the VALU-filter scan admits below an estimate of the same kind: enough for k, ~2 k each
    assert 1024 <= p1["n_candidates"] <= 4 * 1024 and 1024 <= p2["n_candidates"] <= 4 * 1024

    # batched queries: both scans admit below an ESTIMATE of the k-th smallest (~2 k windows expected below it; enough for k or
    # the status says so) -- the matrix-core bootstrap's minima are upper bounds, the VALU one's exact: the result is the same
    q4 = syn.rolling_queries(6, 20, 79)
    d3, i3, s3, p3 = hip_scan(hip_device, ds, q4, 1024, 20, profile=True)
    d4, i4, s4, p4 = hip_scan(hip_device, ds, q4, 1024, 20, profile=True, flags=_native.FLAG_FILTER_VALU)
    assert not s3.any() and not s4.any()
    assert_exact(d3, i3, d4, i4, "mq vs valu filter")
    assert 1024 <= p3["n_candidates"] <= 4 * 1024 and 1024 <= p4["n_candidates"] <= 4 * 1024    # both admit ~2 k (estimates of the same level)


@pytest.mark.parametrize("kind", ["spikes", "tiny_query", "planted", "one_loud_row", "f16_overflow_inf", "zero_rows"])
def test_batched_matrix_core_scan_is_exact_on_adversarial_data(hip_device, oracle_mod, kind):
    """scan_mq_kernel / boot_mq_kernel (4 queries x 8 shifts per MFMA): same guarantee for a batch."""
    R, T, h, k, B = 5000, 2048, 9, 300, 7                        # B = 7: a ragged last query group
    ds, q1 = _adversarial(kind, R, T, 950 + MX_KINDS.index(kind))
    q = np.concatenate([q1, syn.gbm_log_returns((B - 1, 20), 960) * np.float32(q1.std() / 0.0126 + 1e-30)], 0)
    if kind == "planted":
        q[3] = ds[11, 500:520]                                    # one query is an exact window of the data
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, h, profile=True)
    assert prof["path"] == 0
    bad = np.nonzero(status)[0]
    if bad.size:
        d2, idx2, _, _ = hip_scan(hip_device, ds, q[bad], k, h, exhaustive=True)
        d[bad], idx[bad] = d2, idx2
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, kind)


def test_batched_scan_with_more_survivors_than_the_queue_holds(hip_device, oracle_mod):
    """Every row identical: every window position is a 3000-fold tie, the survivor queue of a query
    group overflows and the group is redone with the exact chain (or the slices overflow and the host
    reruns exhaustively): results stay exact."""
    row = syn.dataset(1, 1500, 970)[0, 0]
    ds = np.tile(row, (3000, 1))
    q = np.stack([row[100:120], row[700:720], syn.gbm_log_returns((20,), 971)]).astype(np.float32)
    d, idx, status, _ = hip_scan(hip_device, ds, q, 500, 5)
    bad = np.nonzero(status)[0]
    if bad.size:
        d2, idx2, _, _ = hip_scan(hip_device, ds, q[bad], 500, 5, exhaustive=True)
        d[bad], idx[bad] = d2, idx2
    od, oidx = oracle_mod.scan_topk(ds, q, 500, h=5)
    assert_exact(d, idx, od, oidx, "identical rows")


@pytest.mark.parametrize("W,B", [(1, 1), (8, 1), (17, 1), (24, 1), (33, 1), (34, 1), (8, 3), (24, 5), (25, 6), (26, 2)])
def test_matrix_core_kernels_at_run_time_window_lengths(hip_device, oracle_mod, W, B):
    """W <= 33 (one query) and W <= 25 (batches) take the matrix-core kernels compiled for a run-time W; with per-stage
    profiling (the separate launches) W = 34 / (26, batch) are the first lengths on the VALU test -- one query with a longer
    window otherwise runs stream_scan_long_kernel (test_long_identity_windows_on_the_matrix_cores)."""
    R, T, h, k = 5000, 1100, 7, 200
    ds = syn.dataset(R, T, 1200 + W)
    q = syn.gbm_log_returns((B, W), 1300 + W)
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, h, profile=True)
    assert prof["path"] == 0 and not status.any()
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, f"W={W} B={B}")


@pytest.mark.parametrize("W,R,T,h,k", [(34, 4096, 2048, 7, 200), (50, 4096, 2048, 0, 300), (64, 8192, 2048, 20, 1024), (126, 4096, 2048, 20, 256),
                                       (126, 3000, 2051, 60, 300), (252, 4096, 2048, 20, 200), (256, 2048, 3000, 0, 100), (100, 6000, 1500, 11, 64)])
def test_long_identity_windows_on_the_matrix_cores(hip_device, oracle_mod, W, R, T, h, k):
    """One query with a window of 34 .. 256 samples (the reference takes any Identity(dimension), path_embedding.py:135-139; the
    tutorial's context is 126): the overlap-friendly launches with the scan's banded product as a K-loop over ceil((W + 31) / 16)
    steps (stream_scan_long_kernel) -- path 3 with or without PSH_FLAG_OVERLAP, status protocol as for short windows, results
    bit for bit the oracle's.  (Rounds 1-4 had only the vector-ALU filter for these lengths.)"""
    from shadowing_amd import _native
    ds = syn.dataset(R, T, 1600 + W)
    q = syn.gbm_log_returns((1, W), 1700 + W)
    ds_t = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :])).to(hip_device)
    q_t = torch.as_tensor(q).to(hip_device)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    for flags in (0, _native.FLAG_OVERLAP):
        info = {}
        ws = _native.Workspace(hip_device)
        d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws, flags=flags, info=info)
        torch.cuda.synchronize()
        assert info["path"] == 3, info
        assert int(st[0]) == 0, f"W={W}: status {int(st[0])}"
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long window W={W} flags={flags}")
    # the checked call (what PathShadowing uses) and the vector-ALU filter on the same inputs
    d, idx = _native.scan_topk_checked(ds_t, q_t, k, h=h)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long window W={W} checked")
    d, idx = _native.scan_topk_checked(ds_t, q_t, k, h=h, flags=_native.FLAG_FILTER_VALU)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long window W={W} VALU filter")


def test_long_window_shadow_with_paths_and_adversarial_data(hip_device, oracle_mod):
    """shadow(cuda=True) with Identity(126) (paths included), and the long-window scan on spikes / planted matches / a quiet
    ensemble through the status protocol."""
    import shadowing_amd as sa
    from _adversarial import make
    W, h = 126, 60
    ds = syn.dataset(4096, 2048, 1800)
    obj = sa.PathShadowing(sa.Identity(W), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(horizon=h))
    q = syn.gbm_log_returns((W,), 1801)
    d, paths, idx = obj.shadow(q, k=300, cuda=True)
    od, opaths, oidx = oracle_mod.shadow(ds, q[None, :], 300, h)
    assert_exact(d, idx, od, oidx, "Identity(126) shadow")
    assert np.array_equal(paths, opaths) and obj.last_path == "hip"
    # several query dates with the long context in ONE call (the loop of one-query steps inside psh_scan_topk)
    qb = syn.rolling_queries(5, W, 1802)
    d, paths, idx = obj.shadow(qb, k=200, cuda=True)
    od, opaths, oidx = oracle_mod.shadow(ds, qb, 200, h)
    assert_exact(d, idx, od, oidx, "Identity(126) shadow, 5 queries")
    assert np.array_equal(paths, opaths) and obj.last_path == "hip"
    from shadowing_amd import _native
    for kind in ("spikes", "planted_matches", "scale_down", "quiet_stretches", "zero_constant_rows", "huge_queries"):
        dsa, qa = make(kind, 2048, 2048, 1, 64, 5, 1900)
        d, idx = _native.scan_topk_checked(torch.as_tensor(dsa).to(hip_device), torch.as_tensor(qa).to(hip_device), 200, h=5)
        od, oidx = oracle_mod.scan_topk(dsa, qa, 200, h=5)
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long window, {kind}")


@pytest.mark.parametrize("W", [40, 66, 97, 129, 145, 161, 162, 200, 256])
@pytest.mark.parametrize("B", [1, 2, 3])
def test_long_window_sample_on_the_matrix_cores(hip_device, oracle_mod, W, B):
    """The long-window step's sample as matrix-core UPPER bounds of the unit minima (stream_sample_long_kernel, round 6): every
    K-step bucket (6 / 10 / 14 / 18), one query (one-wave blocks) and two / three (four-wave blocks; beyond what rides one pass a
    loop of steps), a T that leaves a ragged last segment, a horizon -- status OK from the three launches themselves (the level
    held: at least k windows below it) and the oracle's results; then the same ensemble with NaN / inf samples through
    shadow(cuda=True)."""
    from shadowing_amd import _native
    R, T, h, k = 3072, 2300, 9, 150
    ds = syn.dataset(R, T, 4100 + W)
    q = syn.gbm_log_returns((B, W), 4200 + W + B)
    ds_t = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :])).to(hip_device)
    q_t = torch.as_tensor(q).to(hip_device)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    info = {}
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, info=info)
    torch.cuda.synchronize()
    # (path 3: the three launches; two / three queries whose tables do not fit beside the scan's rows -- W > 97 / 145 -- take
    #  the batched long-window scan through the separate launches, path 0, whose BOOT pass is the same construction)
    assert info["path"] == (3 if B == 1 or W <= 97 else info["path"]) and info["path"] in (0, 3) and not st.cpu().numpy().any(), (W, B, info, st.tolist())
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long-window sample W={W} B={B}")
    import shadowing_amd as sa
    dirty = ds.copy()
    rng = np.random.default_rng(4300 + W)
    for r in rng.integers(0, R, 40):
        dirty[r, 0, rng.integers(0, T)] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
    obj = sa.PathShadowing(sa.Identity(W), sa.RelativeMSE(), torch.as_tensor(dirty), sa.PredictionContext(h), cache=True)
    d, paths, idx = obj.shadow(q, k=k, cuda=True)
    assert obj.last_path == "hip"
    od, opaths, oidx = oracle_mod.shadow(dirty, q, k, h)
    assert_exact(d, idx, od, oidx, f"long-window sample, dirty ensemble W={W} B={B}")
    assert np.array_equal(paths, opaths, equal_nan=True)


@pytest.mark.parametrize("W,flags", [(20, "overlap"), (33, "overlap"), (40, 0), (64, 0), (126, 0), (126, "overlap"), (250, 0)])
def test_smooth_ensembles_fill_block_lists_without_giving_up(hip_device, oracle_mod, W, flags):
    """A smooth ensemble (price LEVELS: random walks, not returns) puts a match's neighbours in t next to it in distance and
    makes the f16 test's slack -- proportional to ||x|| ||y|| -- large beside the admission level: thousands of admitted
    windows in a few blocks.  A block's 64-entry list then overflows into the query's list in memory (spill_candidate) instead
    of failing the step: status OK on the three launches, results the oracle's.  (Until round 5 such a step answered
    PSH_STATUS_RETRY -- correct through the protocol, but three passes over the ensemble.)"""
    from shadowing_amd import _native
    R, T, h, k = 8192, 2048, 0, 1024
    rng = np.random.default_rng(2100 + W)
    ds = (0.05 * np.cumsum(rng.standard_normal((R, T)), axis=1)).astype(np.float32)
    q = (0.05 * np.cumsum(rng.standard_normal((1, W)))).astype(np.float32).reshape(1, W)
    ds_t, q_t = torch.as_tensor(ds).to(hip_device), torch.as_tensor(q).to(hip_device)
    info = {}
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, flags=_native.FLAG_OVERLAP if flags else 0, info=info)
    torch.cuda.synchronize()
    assert info["path"] == 3 and int(st[0]) == 0, (info, st.tolist())
    od, oidx = oracle_mod.scan_topk(ds[:, None, :], q, k, h=h)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"smooth ensemble W={W}")
    # two and three queries ride the same launches (W <= 33)
    if W <= 33:
        q3 = np.concatenate([q, q[:, ::-1], 0.5 * q + 0.01], axis=0).astype(np.float32)
        d, idx, st = _native.scan_topk(ds_t, torch.as_tensor(q3).to(hip_device), k, h=h, info=info)
        torch.cuda.synchronize()
        assert info["path"] == 3 and not st.cpu().numpy().any(), (info, st.tolist())
        od, oidx = oracle_mod.scan_topk(ds[:, None, :], q3, k, h=h)
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"smooth ensemble W={W}, three queries")


@pytest.mark.parametrize("W,B,h,k", [(34, 2, 3, 100), (64, 5, 20, 300), (126, 4, 60, 256), (126, 17, 0, 64), (252, 3, 20, 128),
                                     (26, 10, 3, 64), (30, 7, 5, 128), (33, 4, 0, 200)])
def test_long_window_batches_loop_over_the_one_query_step(hip_device, oracle_mod, W, B, h, k):
    """Several queries with a window of 34 .. 256 samples: psh_scan_topk serves them as B one-query steps (path 3, the
    matrix-core long-window scan) -- the batched kernels' bands stop at W = 25 -- with per-query status words, a hint per
    query, and the same results as the one-pass vector-ALU scan (PSH_FLAG_FILTER_VALU) and the oracle.  Four queries and
    more with 26 <= W <= 33: three queries per step."""
    from shadowing_amd import _native
    R, T = 4096, 2048
    ds = syn.dataset(R, T, 2200 + W)
    q = syn.gbm_log_returns((B, W), 2300 + W + B)
    ds_t = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :])).to(hip_device)
    q_t = torch.as_tensor(q).to(hip_device)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    info = {}
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, info=info)
    torch.cuda.synchronize()
    # round 6: four queries and more with W >= 34 are ONE pass per chunk of queries of the batched long-window scan (psh_lq.hip,
    # through the separate launches: path 0); fewer queries, 26 <= W <= 33 and PSH_FLAG_LONG_LOOP keep the loop of steps
    # (the path of its LAST step: the three launches, or -- one query left over with W <= 33 -- the fused launch)
    # (later in round 6 also two / three queries that do not ride one pass of the three launches: three up to W = 97, two up to 145)
    # (... and batches of four and more with 26 <= W <= 33, until then a loop of three-query steps)
    batched = (W > 33 and (B >= 4 or B > (3 if W <= 97 else (2 if W <= 145 else 1)))) or (W >= 26 and B >= 4)
    assert info["path"] == (0 if batched else (3 if W > 33 or B % 3 != 1 else 2)), info
    stn = st.cpu().numpy()
    if batched:
        # (the separate launches admit below a sampled estimate per query: on an ensemble this small a query's estimate may fall
        #  short of k windows -- its status says so, PSH_STATUS_OVERFLOW, and the checked call reruns that query: the protocol)
        ok = stn == 0
        assert ok.sum() >= (B + 1) // 2, stn
        assert_exact(d.cpu().numpy()[ok], idx.cpu().numpy()[ok], od[ok], oidx[ok], f"long-window batch W={W} B={B}: the queries that report OK")
        d, idx = _native.scan_topk_checked(ds_t, q_t, k, h=h)
    else:
        assert not stn.any(), (info, stn)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long-window batch W={W} B={B}")
    if batched:
        info = {}
        d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, info=info, flags=_native.FLAG_LONG_LOOP)
        torch.cuda.synchronize()
        assert info["path"] == (3 if W > 33 or B % 3 != 1 else 2) and not st.cpu().numpy().any(), (info, st.tolist())
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long-window batch W={W} B={B}, the loop of steps")
    info = {}
    _native.scan_topk(ds_t, q_t, k, h=h, info=info, flags=_native.FLAG_FILTER_VALU)
    assert info["path"] == 0, info
    d, idx = _native.scan_topk_checked(ds_t, q_t, k, h=h, flags=_native.FLAG_FILTER_VALU)      # (its per-query OVERFLOW is part of the protocol)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long-window batch W={W} B={B}, one pass")
    # a level per query: acc of the k-th window x 1.1 (OK), and one query's hint far too low (that query alone reports it)
    xn2 = (q.astype(np.float64) ** 2).sum(axis=1)
    lev = ((od[:, k - 1].astype(np.float64) ** 2) * xn2 * 1.1).astype(np.float32)
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, tau_hint=torch.as_tensor(lev).to(hip_device))
    torch.cuda.synchronize()
    assert not st.cpu().numpy().any()
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long-window batch W={W} B={B}, hints")
    lev[B - 1] *= 1e-3
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, tau_hint=torch.as_tensor(lev).to(hip_device))
    torch.cuda.synchronize()
    stn = st.cpu().numpy()
    assert stn[B - 1] != 0 and not stn[: B - 1].any(), stn
    if not batched:                                    # (the three launches poison a query's invalid results; the separate launches' status says it)
        assert np.isnan(d[B - 1].cpu().numpy()).all()
    assert_exact(d[: B - 1].cpu().numpy(), idx[: B - 1].cpu().numpy(), od[: B - 1], oidx[: B - 1], "the other queries of the call")
    d, idx = _native.scan_topk_checked(ds_t, q_t, k, h=h, tau_hint=torch.as_tensor(lev).to(hip_device))
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"long-window batch W={W} B={B}, checked with a short hint")


@pytest.mark.parametrize("B,W,r_offset,dup", [(1, 20, 0, False), (3, 20, 0, False), (1, 64, 0, False), (1, 20, 1 << 22, False), (2, 24, 0, True)])
def test_many_admitted_windows_are_ranked_below_a_level(hip_device, oracle_mod, B, W, r_offset, dup):
    """Tens of thousands of admitted windows per query (a generous hint here; a smooth ensemble in practice) through the ranking
    launch of the three launches: lists far beyond a block's 64 entries and the header's 16384, the k best in (d, r, t) order,
    packed and unpacked keys, ties everywhere (duplicated rows).  (A ranking that first finds a level holding the k best and
    ranks only the candidates below it -- three counting passes on the distance bits -- was built for this case in round 5 and
    was no faster: a pass by one wave is bound by the latency of its loads, not by the entries it moves; DESIGN_APPENDIX A.5.)"""
    from shadowing_amd import _native
    R, T, h, k, m = 6000, 2048, 5, 1024, 30000
    ds = syn.dataset(R, T, 2500 + W + B)
    if dup:
        ds[1::2] = ds[0::2]                                    # every distance twice: ties everywhere, also at the level's edge
    q = syn.gbm_log_returns((B, W), 2600 + W + B)
    od, oidx = oracle_mod.scan_topk(ds, q, m, h=h, r_offset=r_offset) if r_offset else oracle_mod.scan_topk(ds, q, m, h=h)
    lev = ((od[:, m - 1].astype(np.float64) ** 2) * (q.astype(np.float64) ** 2).sum(axis=1) * (1.0 + 1e-6)).astype(np.float32)
    ds_t = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :])).to(hip_device)
    q_t = torch.as_tensor(q).to(hip_device)
    ws = _native.Workspace(hip_device)
    info = {}
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, r_offset=r_offset, workspace=ws, flags=_native.FLAG_OVERLAP, info=info,
                                   tau_hint=torch.as_tensor(lev).to(hip_device))
    torch.cuda.synchronize()
    assert info["path"] == 3 and not st.cpu().numpy().any(), (info, st.tolist())
    lay = _native.candidates_layout(R, T, B, W, h, k, ws.buf.numel())
    ncand = ws.buf[lay["hdr_stream_ncand"]: lay["hdr_stream_ncand"] + 4 * B].view(torch.int32).cpu().numpy()
    assert (ncand > 8192).all() and (ncand >= m - 64).all(), ncand       # the mode under test; ~m windows lie below the level
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od[:, :k], oidx[:, :k], f"many candidates B={B} W={W} r_offset={r_offset}")


def test_shape_fuzz_cut(hip_device):
    """A cut of tests/stress/stress_shapes.py (the whole script: 1300 cases on an MI355X in round 5, no mismatch): random window
    lengths 1 .. 256, 1 .. 20 queries, ragged rows, adversarial and smooth ensembles, both flag settings, good and short
    hints, through the status protocol -- whatever launch structure the library picks -- against the oracle, bit for bit."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("stress_shapes", Path(__file__).parent / "stress" / "stress_shapes.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, seen = mod.run(11, 70, verbose=False)
    assert bad == 0
    assert {0, 2, 3} <= set(seen), seen                     # separate launches, the fused launch, the three launches: all taken


def test_unsorted_flag_returns_the_same_set(hip_device, oracle_mod):
    """PSH_FLAG_UNSORTED (what the sharded scan asks of its local selection): the k best, any order."""
    ds = syn.dataset(4096, 2048, 1400)
    q = syn.rolling_queries(3, 20, 1401)
    d, idx, status, _ = hip_scan(hip_device, ds, q, 512, 20, unsorted=True)
    assert not status.any()
    od, oidx = oracle_mod.scan_topk(ds, q, 512, h=20)
    dc, ic = canonical(d, idx)
    assert_exact(dc, ic, od, oidx, "unsorted flag")


def test_two_class_slices_fall_back_when_the_estimate_is_too_low(hip_device, oracle_mod):
    """The single-query scan files candidates below tau2 -- an ESTIMATE of the k-th smallest acc from the
    bootstrap rows -- at the front of each block's slice and the other admitted ones at the back.  Near-copies
    of the query planted in the bootstrap rows ONLY drag the estimate far below the true k-th value: fewer
    than k candidates sit in front, and the selection must take the back lists too."""
    R, T, h, k = 8192, 2048, 20, 1024
    ds = syn.dataset(R, T, 1500)[:, 0, :].copy()
    q = syn.single_query(20, 1501)[None, :]
    rng = np.random.default_rng(1502)
    for i in range(200):                                   # rows 8, 24, 40, ...: the ones the bootstrap visits
        r = 8 + 16 * int(rng.integers(0, R // 16))
        t = int(rng.integers(0, T - 60))
        ds[r, t:t + 20] = q[0] * (1 + 0.02 * rng.standard_normal(20).astype(np.float32))
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, h, profile=True)
    assert prof["path"] == 0 and status[0] == 0
    assert prof["n_candidates"] >= k                       # front lists alone (~200 planted) could not have supplied k
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, "two-class fallback")
    # and the normal case on the same sizes: the front lists suffice and are a fraction of what tau admits
    ds2 = syn.dataset(R, T, 1503)
    d2, idx2, st2, prof2 = hip_scan(hip_device, ds2, q, k, h, profile=True)
    assert st2[0] == 0 and k <= prof2["n_candidates"] < 6 * k
    od2, oidx2 = oracle_mod.scan_topk(ds2, q, k, h=h)
    assert_exact(d2, idx2, od2, oidx2, "two-class normal")


@pytest.mark.parametrize("k", [4096, 8192, 16384])
def test_single_query_with_a_large_k_admits_below_an_estimate(hip_device, oracle_mod, k):
    """The reference's own example call (testing.ipynb: Identity(20), R = 32768, k = 8192): the provable bound would want 8 k
    segment minima -- more than the rows have to give -- so the scan admits below an ESTIMATE from a 1/32 sample (~1.5 k windows
    expected below it) and the selection checks that k were found.  Exact against the oracle; the candidates stay a small
    multiple of k.  Near-copies of the query planted in sampled rows only drag the estimate down: fewer than k found -> status
    -> the exhaustive pass, still exact."""
    from shadowing_amd import _native
    R, T, h = 32768, 4096, 20
    ds = syn.dataset(R, T, 0)[:, 0, :]
    q = syn.single_query(20, 1801)[None, :]
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, h, profile=True)
    assert status[0] == 0 and prof["path"] == 0 and prof["n_sample_rows"] == 1024
    assert k <= prof["n_candidates"] < 3 * k
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, f"large k = {k}")
    if k == 8192:
        ds2 = ds.copy()
        rng = np.random.default_rng(1802)
        for r in range(16, R, 32):                             # the sampled rows (stride 32, first row 16): a near-copy in each
            t = int(rng.integers(0, T - 60))
            ds2[r, t:t + 20] = q[0] * (1 + 0.01 * rng.standard_normal(20).astype(np.float32))
        dev = torch.device("cuda", 0)
        dsd, qd = torch.as_tensor(ds2).to(dev), torch.as_tensor(q).to(dev)
        _, _, st = _native.scan_topk(dsd, qd, k, h=h)
        assert int(st[0]) != 0                                 # the estimate sits among the planted copies: short of k
        d2, i2 = _native.scan_topk_checked(dsd, qd, k, h=h)
        od2, oidx2 = oracle_mod.scan_topk(ds2, q, k, h=h)
        assert_exact(d2.cpu().numpy(), i2.cpu().numpy(), od2, oidx2, "large k, estimate fooled")


@pytest.mark.parametrize("G,k_in,k,B", [(8, 1024, 1024, 1), (3, 500, 700, 2), (2, 64, 200, 4), (1, 128, 128, 2), (5, 100, 37, 2)])
def test_merge_of_sorted_gathered_lists(hip_device, oracle_mod, G, k_in, k, B):
    """psh_merge_sorted_gathered: G sorted per-shard lists (ascending row blocks) -> global k best, positions
    by binary search.  Includes k larger than one list, k larger than everything (padding), ties across
    shards (duplicated rows) and a shard shorter than k_in (padded with +inf / -1)."""
    from shadowing_amd import _native
    rows, T, h = 300, 400, 5
    base = syn.dataset(rows, T, 1600 + G)
    ds = np.concatenate([base] * G, 0) if G in (3, 5) else syn.dataset(rows * G, T, 1601 + G)   # duplicated shards: exact ties
    q = syn.gbm_log_returns((B, 20), 1602)
    gathered = torch.empty((G, 3 * B * k_in), dtype=torch.int32, device=hip_device)
    for g in range(G):
        lo, hi = g * rows, (g + 1) * rows
        if g == G - 1 and G == 2:
            hi = lo + 1                                   # a one-row shard: fewer than k_in windows? (376 windows: no) keep it small anyway
        od, oi = oracle_mod.scan_topk(ds[lo:hi], q, min(k_in, (hi - lo) * (T - 20 - h + 1)), h=h, r_offset=lo)
        dd = np.full((B, k_in), np.inf, np.float32); ii = np.full((B, k_in, 2), -1, np.int32)
        dd[:, :od.shape[1]] = od; ii[:, :oi.shape[1]] = oi
        gathered[g, :B * k_in] = torch.from_numpy(dd.reshape(-1).view(np.int32)).to(hip_device)
        gathered[g, B * k_in:] = torch.from_numpy(ii.reshape(-1)).to(hip_device)
    md, mi = _native.merge_sorted_gathered(gathered, G, B, k_in, k)
    torch.cuda.synchronize()
    n_rows = (G - 1) * rows + (1 if G == 2 else rows)
    n_all = n_rows * (T - 20 - h + 1)
    od, oidx = oracle_mod.scan_topk(ds[:n_rows], q, min(k, n_all, G * k_in), h=h)
    kk = od.shape[1]
    # the merged list can only be exact where every shard contributed enough: compare the guaranteed prefix
    sure = min(kk, k_in)
    assert_exact(md.cpu().numpy()[:, :sure], mi.cpu().numpy()[:, :sure], od[:, :sure], oidx[:, :sure], f"sorted merge G={G}")
    gd, gi = _native.merge_topk_gathered(gathered, G, B, k_in, k)
    real = gi.cpu().numpy()[..., 0] >= 0
    assert np.array_equal(md.cpu().numpy()[real].view(np.uint32), gd.cpu().numpy()[real].view(np.uint32))
    assert np.array_equal(mi.cpu().numpy()[real], gi.cpu().numpy()[real])


@pytest.mark.parametrize("name", ["forward_topk_d34", "forward_topk_d5", "forward_topk_d126"])
def test_forward_topk_on_the_device_matches_reference_golden(hip_device, name):
    """RelativeMSE.forward_topk with device tensors runs the scan kernels (N one-window paths): the reference's
    CPU output bit for bit, int64 indices into y's leading dims."""
    from pathlib import Path
    import shadowing_amd as sa
    z = np.load(Path(__file__).resolve().parent / "golden" / f"{name}.npz")
    x, y, k = torch.tensor(z["x"]).to(hip_device), torch.tensor(z["y"]).to(hip_device), int(z["k"])
    d, idx = sa.RelativeMSE().forward_topk(x, y, k, n_splits=int(z["n_splits"]))
    assert d.is_cuda and idx.dtype == torch.int64 and tuple(idx.shape) == z["idx"].shape
    assert np.array_equal(d.cpu().numpy().view(np.uint32), np.sort(z["d"], 1).view(np.uint32))
    for b in range(x.shape[0]):
        assert {tuple(v) for v in idx[b].cpu().numpy()} == {tuple(v) for v in z["idx"][b]}


def test_forward_topk_on_the_device_large(hip_device, oracle_mod):
    """A pre-embedded ensemble of 2^18 points (4-D y): sampled path sizes, against the oracle."""
    g = np.random.default_rng(8)
    y = (g.standard_normal((64, 128, 32, 34)) * 0.02).astype(np.float32)
    x = (g.standard_normal((3, 34)) * 0.02).astype(np.float32)
    import shadowing_amd as sa
    d, idx = sa.RelativeMSE().forward_topk(torch.tensor(x).to(hip_device), torch.tensor(y).to(hip_device), 200)
    od, oi = oracle_mod.scan_topk(y.reshape(-1, 1, 34), x, 200, h=0)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))
    flat = (idx[..., 0] * 128 + idx[..., 1]) * 32 + idx[..., 2]
    assert np.array_equal(flat.cpu().numpy(), oi[..., 0])


@pytest.mark.parametrize("k,B", [(12000, 1), (6000, 3), (4096, 2), (3000, 2)])
def test_large_k_ordering_with_duplicated_paths(hip_device, oracle_mod, k, B):
    """Large k over an ensemble whose rows repeat 8 times: every distance value occurs 8 times, so the ordering stage
    (kpad >= 4096: 1024-item chunk sorts and a merge by ranking over kpad / 1024 blocks per query; below: one block's merge
    sort) leans on the (r, t) tie-break everywhere -- inside a wave's run, across runs, across chunks."""
    base = syn.dataset(256, 1024, 91)
    ds = np.ascontiguousarray(np.tile(base, (8, 1, 1)))
    q = syn.gbm_log_returns((B, 20), 92)
    d, idx, status, _ = hip_scan(hip_device, ds, q, k, 20)
    if np.any(status != 0):
        d, idx, _, _ = hip_scan(hip_device, ds, q, k, 20, exhaustive=True)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=20)
    assert_exact(d, idx, od, oidx, "large k, 8-fold ties")


@pytest.mark.parametrize("R,W,h,k,B", [
    (5000, 20, 3, 100, 2),      # T = 23: flat staging (odd row length)
    (3000, 32, 0, 64, 1),       # T = 32: one row/column split per float4, odd LDS row stride
    (4097, 33, 7, 300, 3),      # T = 40, W = 33: quads with unused tail columns; ragged last chunk
    (2000, 256, 0, 50, 1),      # PSH_MAX_W: 132 KB of LDS per block
    (6000, 5, 100, 200, 2),     # T = 105 >> W: per-element split, only the first W samples kept
    (70000, 34, 0, 1000, 9),    # the sampled path at size, 9 queries
    (1000, 7, 0, 900, 1),       # k close to N / 1: bootstrap too thin -> exhaustive, one slot per row
])
def test_one_window_rows_equal_oracle(hip_device, oracle_mod, R, W, h, k, B):
    """Paths exactly one window long (T == W + h): rows_kernel (a row per lane; BOOT / FILTER / ALL and its three
    staging modes) against the oracle's contiguous 8-lane reduce -- what forward_topk and shadow() on such paths run."""
    ds = syn.dataset(R, W + h, 300 + R)
    q = syn.gbm_log_returns((B, W), 400 + W)
    d, idx, status, _ = hip_scan(hip_device, ds, q, k, h)
    if np.any(status != 0):
        d, idx, _, _ = hip_scan(hip_device, ds, q, k, h, exhaustive=True)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, f"one-window rows R={R} W={W} h={h} k={k} B={B}")
    assert np.all(idx[..., 1] == 0)


RANK_CASES = [   # (flags, W, B, k, what)
    ("FLAG_NO_FUSE", 20, 1, 1024, "two-class slices, front lists"),
    ("FLAG_FILTER_VALU", 20, 1, 1024, "VALU-filter scan"),
    ("FLAG_FILTER_VALU", 20, 2, 700, "two queries"),
    ("FLAG_FILTER_VALU", 40, 1, 2048, "W = 40"),
]


@pytest.mark.parametrize("flag,W,B,k,what", RANK_CASES)
def test_selection_by_ranking_on_all_cus_equals_the_one_block_selection(hip_device, oracle_mod, flag, W, B, k, what):
    """One or two queries, k <= 4096, at most 8192 candidates: rank_select_kernel (256 blocks per query, each ranking its
    share of the candidates against all of them by counting) writes the results and select_kernel returns at once;
    PSH_FLAG_SELECT_ONE_BLOCK keeps the one-block radix select + sort.  Same bytes, and the oracle's.  The ensemble
    repeats its rows 4 times: every distance occurs 4 times and (r, t) decides."""
    from shadowing_amd import _native
    base = syn.dataset(4096, 2048, 1700)
    ds = np.ascontiguousarray(np.tile(base, (4, 1, 1)))
    q = syn.gbm_log_returns((B, W), 1701)
    fl = getattr(_native, flag)
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, 20, flags=fl, profile=True)
    d1, idx1, status1, prof1 = hip_scan(hip_device, ds, q, k, 20, flags=fl | _native.FLAG_SELECT_ONE_BLOCK, profile=True)
    assert prof["path"] == 0 and not status.any() and not status1.any()
    assert prof["n_candidates"] == prof1["n_candidates"] <= 8192
    assert_exact(d, idx, d1, idx1, what + ": ranking vs one block")
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=20)
    assert_exact(d, idx, od, oidx, what + ": ranking vs oracle")


def test_selection_by_ranking_leaves_what_it_cannot_take(hip_device, oracle_mod):
    """More than 8192 candidates (an estimate dragged up by rows that repeat 16 times), and row indices that do not pack
    with t into 32 bits (r_offset near 2^31 / T'): the ranking kernel declines, the one-block selection does the work."""
    from shadowing_amd import _native
    base = syn.dataset(1024, 2048, 1710)
    ds = np.ascontiguousarray(np.tile(base, (16, 1, 1)))
    q = syn.gbm_log_returns((1, 20), 1711)
    k = 1024
    d, idx, status, prof = hip_scan(hip_device, ds, q, k, 20, flags=_native.FLAG_FILTER_VALU, profile=True)
    if status.any():
        d, idx, _, _ = hip_scan(hip_device, ds, q, k, 20, exhaustive=True)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=20)
    assert_exact(d, idx, od, oidx, "many candidates")
    ds2 = syn.dataset(16384, 2048, 1712)
    off = (1 << 20) + 12345                                 # 2^20 rows x 2^11 windows: one bit too many
    d2, idx2, st2, _ = hip_scan(hip_device, ds2, q, k, 20, flags=_native.FLAG_FILTER_VALU, r_offset=off)
    assert not st2.any()
    od2, oidx2 = oracle_mod.scan_topk(ds2, q, k, h=20)
    oidx2 = oidx2.copy(); oidx2[..., 0] += off
    assert_exact(d2, idx2, od2, oidx2, "row offset beyond the packed key")


@pytest.mark.parametrize("script,seed,cases", [("stress_mx.py", 3, 25), ("stress_emx.py", 5, 25), ("stress.py", 7, 20)])
def test_promoted_cuts_of_the_remaining_stress_scripts(hip_device, script, seed, cases):
    """tests/stress/stress_mx.py (the single-query matrix-core scan on adversarial data), stress_emx.py (the dense embedded scan's
    matrix-core rejection test against the vector-ALU one and the oracle) and stress.py (the separate launches) used to run outside
    pytest only (round 5's verdict): 70 cases of them here -- each script is its own process, as on the command line, and says
    `mismatches: 0`."""
    import subprocess
    import sys
    from pathlib import Path
    res = subprocess.run([sys.executable, str(Path(__file__).parent / "stress" / script), str(seed), str(cases)],
                         capture_output=True, text=True, timeout=1500)
    tail = res.stdout[-2500:]
    assert res.returncode == 0 and "MISMATCH" not in res.stdout and "mismatches: 0" in res.stdout, tail + res.stderr[-1500:]
