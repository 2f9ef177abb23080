"""BASELINE.json configs[2] -- many query dates sharing one pass over the ensemble -- at sizes that run the
batched scan's query chunks for real (scan_mq_kernel / boot_mq_kernel take PSH_MQ_CHUNK = 112 queries per
blockIdx.y; one launch carries at most PSH_MAX_B_PER_LAUNCH = 1024), and the seam itself:
PathShadowing.batched_distance(cuda=True) / shadow(cuda=True) against the reference's own outputs."""
import numpy as np
import pytest
import torch

from _util import SMALL_GOLDENS, assert_exact, assert_matches_reference, bits, load_golden, rows3
from shadowing_amd import synthetic as syn

pytestmark = pytest.mark.gpu

BATCHED_GOLDENS = ["cfg3_rolling_B128_R256", "cfg3_rolling_B130_R1024"]


def hip_scan(dev, ds, q, k, h, **kw):
    from shadowing_amd import _native
    ds_t = torch.as_tensor(np.ascontiguousarray(rows3(ds)[:, 0, :])).to(dev)
    q_t = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(q), dtype=np.float32)).to(dev)
    out = _native.scan_topk(ds_t, q_t, k, h=h, **kw)
    torch.cuda.synchronize(dev)
    return out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy(), out[3:] and out[3]


def resolve(dev, ds, q, k, h, d, idx, status):
    bad = np.nonzero(status)[0]
    if bad.size:
        d2, idx2, _, _ = hip_scan(dev, ds, q[bad], k, h, exhaustive=True)
        d[bad], idx[bad] = d2, idx2
    return d, idx


@pytest.mark.parametrize("name", BATCHED_GOLDENS)
def test_more_queries_than_one_chunk_match_the_reference(hip_device, oracle_mod, name):
    """128 / 130 rolling query dates (two query chunks, the second one ragged): the reference's CPU output."""
    g = load_golden(name)
    d, idx, status, prof = hip_scan(hip_device, g["dataset"], g["queries"], g["k"], g["h"], profile=True)
    if name.endswith("R1024"):       # (R = 256 fits the candidate buffer whole: the exhaustive path, 128 queries wide)
        assert prof["path"] == 0, "the sampled path (bootstrap -> threshold -> batched scan -> select) must be the one tested"
    d, idx = resolve(hip_device, g["dataset"], g["queries"], g["k"], g["h"], d, idx, status)
    assert_matches_reference(d, idx, g, None, what=name)
    od, oidx = oracle_mod.scan_topk(g["dataset"], g["queries"], g["k"], h=g["h"])
    assert_exact(d, idx, od, oidx, name + " vs oracle")


@pytest.mark.parametrize("B,R,T,k", [(113, 2048, 1024, 64), (225, 2048, 1024, 100), (512, 2048, 2048, 256),
                                     (1100, 1536, 1024, 64)])
def test_query_chunks_equal_oracle_seeded(hip_device, oracle_mod, B, R, T, k):
    """B = 113 (one query into the second chunk), 225 (a full + a one-query chunk... of three), 512 (configs[2]'s
    batch), 1100 (the host splits at PSH_MAX_B_PER_LAUNCH = 1024): every query bit-exact against the oracle."""
    ds = syn.dataset(R, T, 2000 + B)
    q = syn.rolling_queries(B, 20, 2100 + B)
    d, idx, status, _ = hip_scan(hip_device, ds, q, k, 20)
    assert status.shape == (B,)
    d, idx = resolve(hip_device, ds, q, k, 20, d, idx, status)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=20)
    assert_exact(d, idx, od, oidx, f"B={B}")


def test_query_chunks_with_independent_queries_and_runtime_window_length(hip_device, oracle_mod):
    """Independent draws instead of rolling windows (no shared samples between neighbours), W = 17 (run-time
    window length), B = 130."""
    ds = syn.dataset(1024, 1500, 2300)
    q = syn.gbm_log_returns((130, 17), 2301)
    q[7] *= 30.0                                              # one loud query in the common f16 scale
    q[120] = ds[5, 0, 200:217]                                # one exact window of the data
    d, idx, status, prof = hip_scan(hip_device, ds, q, 77, 5, profile=True)
    assert prof["path"] == 0
    d, idx = resolve(hip_device, ds, q, 77, 5, d, idx, status)
    od, oidx = oracle_mod.scan_topk(ds, q, 77, h=5)
    assert_exact(d, idx, od, oidx, "independent queries")


def test_8bit_and_f16_rejection_tests_give_the_same_results(hip_device, oracle_mod):
    """The batched scan's rejection test as the 8-bit product (the default) and as the f16 product (PSH_FLAG_MQ_F16): the
    same candidates' exact distances either way, both equal to the oracle."""
    from shadowing_amd import _native
    ds = syn.dataset(2048, 2048, 2600)
    q = syn.rolling_queries(96, 20, 2601)
    d8, i8, s8, p8 = hip_scan(hip_device, ds, q, 200, 20, profile=True)
    d16, i16, s16, p16 = hip_scan(hip_device, ds, q, 200, 20, profile=True, flags=_native.FLAG_MQ_F16)
    assert p8["path"] == 0 and p16["path"] == 0
    d8, i8 = resolve(hip_device, ds, q, 200, 20, d8, i8, s8)
    d16, i16 = resolve(hip_device, ds, q, 200, 20, d16, i16, s16)
    od, oidx = oracle_mod.scan_topk(ds, q, 200, h=20)
    assert_exact(d8, i8, od, oidx, "8-bit test")
    assert_exact(d16, i16, od, oidx, "f16 test")


@pytest.mark.parametrize("case", ["amplitudes", "outliers", "quiet_segments", "nonfinite", "zero_query", "wide_window"])
def test_8bit_rejection_test_on_adversarial_batches(hip_device, oracle_mod, case):
    """What the quantisation bound has to survive: queries whose amplitudes differ by orders of magnitude on ONE step (a quiet
    query keeps more windows for its exact recheck -- slower, never wrong), segments dominated by an outlier (a coarse
    step for everything else in them), segments far below the batch's scale (the step's floor), all-zero rows, NaN / inf in
    the data and in a query, an all-zero query, a 25-sample window (the last length the K = 32 band takes)."""
    W, h, k = 20, 20, 150
    ds = syn.dataset(1536, 1400, 2700)
    q = syn.rolling_queries(40, 20, 2701)
    g = np.random.default_rng(2702)
    if case == "amplitudes":
        q = q * (10.0 ** g.uniform(-3, 3, size=(40, 1))).astype(np.float32)
    elif case == "outliers":
        rows = g.integers(0, 1536, 300)
        ds[rows, 0, g.integers(0, 1400, 300)] *= g.choice([40.0, -300.0, 5000.0], 300).astype(np.float32)
    elif case == "quiet_segments":
        ds[:200] *= 1e-4
        ds[200:230] = 0.0
        ds[230:260, 0, :700] *= 1e-7
        ds[260:270, 0, :] = ds[260:270, 0, :1] + 0.0           # constant rows
    elif case == "nonfinite":
        ds[g.integers(0, 1536, 40), 0, g.integers(0, 1400, 40)] = np.nan
        ds[g.integers(0, 1536, 40), 0, g.integers(0, 1400, 40)] = np.inf
        ds[g.integers(0, 1536, 10), 0, g.integers(0, 1400, 10)] = -np.inf
        q[3, 5] = np.nan
        q[9, 0] = np.inf
    elif case == "zero_query":
        q[11] = 0.0
        q[12] = 1e-30
    elif case == "wide_window":
        W, h = 25, 3
        q = syn.gbm_log_returns((40, 25), 2703)
    ds_scan = ds
    if case == "nonfinite":
        # (the reference's rule for non-finite samples -- its zero-padded conv makes a window NaN when its horizon holds one
        #  too -- is applied to the rows by the library's own kernel, as PathShadowing does before it scans)
        from shadowing_amd import _native
        ds_scan = _native.smear_nonfinite(torch.as_tensor(ds).to(hip_device), back=h).cpu().numpy()[:, None, :]
    d, idx, status, prof = hip_scan(hip_device, ds_scan, q, k, h, profile=True)
    assert prof["path"] == 0
    d, idx = resolve(hip_device, ds_scan, q, k, h, d, idx, status)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    fin = np.isfinite(od).all(axis=1) & np.isfinite(q).all(axis=1)
    assert fin.sum() >= 35
    assert_exact(d[fin], idx[fin], od[fin], oidx[fin], case)
    for b in np.nonzero(~fin)[0]:                                  # a non-finite or all-zero query: nothing finite may come back
        assert not np.isfinite(d[b]).any() and not np.isfinite(od[b]).any()


@pytest.mark.parametrize("kind", ["spikes", "tiny_queries", "huge_queries", "scale_up", "scale_down", "planted_matches", "student_t",
                                  "zero_constant_rows", "loud_rows", "quiet_one_loud", "spread_amplitudes", "quiet_stretches"])
def test_promoted_stress_cut_of_the_8bit_scan(hip_device, oracle_mod, kind):
    """One seed x twelve kinds of tests/stress/stress_mq8.py (the 4 x 36-case stress of round 4 runs outside pytest): the
    8-bit batched scan against the oracle on tests/_adversarial.py's batches, every third kind also through the f16 test."""
    from _adversarial import KINDS, make
    from shadowing_amd import _native
    i = KINDS.index(kind)
    R, T = (2048, 3000, 4096)[i % 3], (768, 1024, 1500)[(i // 3) % 3]
    W, h, k, B = (20, 8, 13, 25, 17, 20)[i % 6], (0, 7, 20, 29)[i % 4], (1, 32, 200, 1024)[i % 4], (32, 33, 48, 64, 100, 130)[i % 6]
    ds, q = make(kind, R, T, B, W, h, 9000 + i)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    for flags in ((0, _native.FLAG_MQ_F16) if i % 3 == 0 else (0,)):
        d, idx, status, prof = hip_scan(hip_device, ds, q, k, h, profile=True, flags=flags)
        assert prof["path"] == 0
        d, idx = resolve(hip_device, ds, q, k, h, d, idx, status)
        assert_exact(d, idx, od, oidx, f"{kind} flags={flags}")


def test_configs2_full_size_properties(hip_device, oracle_mod):
    """BASELINE configs[2] at its size: 512 rolling query dates x R = 32768 x T = 4096, k = 1024.  Size-independent
    properties for every query (rows sorted, indices admissible and distinct, each returned distance re-derived
    on the host in the reference's order for sampled rows) and the full oracle scan for a few queries spread over
    the chunks (first, last of chunk 0, first of chunk 1, the ragged last chunk)."""
    g = load_golden("cfg2_R32768")                           # the ensemble of configs[1]/[2] (SHA-256 checked)
    ds, h, W, k = g["dataset"], g["h"], g["W"], g["k"]
    B = 512
    q = syn.rolling_queries(B, W, syn.QUERY_SEED)
    d, idx, status, _ = hip_scan(hip_device, ds, q, k, h)
    assert not status.any()
    assert np.all(np.diff(d, axis=1) >= 0)
    assert idx[..., 0].min() >= 0 and idx[..., 0].max() < ds.shape[0]
    assert idx[..., 1].min() >= 0 and idx[..., 1].max() <= ds.shape[-1] - W - h
    flat = idx[..., 0].astype(np.int64) * ds.shape[-1] + idx[..., 1]
    assert all(len(np.unique(flat[b])) == k for b in range(B))
    xn = oracle_mod.qnorm(q)
    for b in range(0, B, 7):
        for j in (0, 1, 511, 1023):
            r, t = idx[b, j]
            acc = np.float32(0)
            for i in range(W):
                D = np.float32(q[b, i] - ds[r, 0, t + i])
                acc = np.float32(np.float64(D) * np.float64(D) + np.float64(acc))
            assert bits(np.float32(np.sqrt(acc)) / xn[b]) == bits(d[b, j]), (b, j)
    for b in (0, 111, 112, 300, 511):
        od, oidx = oracle_mod.scan_topk(ds, q[b:b + 1], k, h=h)
        assert_exact(d[b:b + 1], idx[b:b + 1], od, oidx, f"configs[2] query {b}")


# ---- the seam: batched_distance / shadow through the reference's API ------------------------------
@pytest.mark.parametrize("name", ["cfg1_h20", "cfg1_hNone", "multiquery_splits", "remainder_split_W12", "oddW33_h11",
                                  "W7_h0", "self_match", "single_window_rows", "cfg3_rolling_B128_R256"])
def test_batched_distance_cuda_matches_the_reference(hip_device, name):
    """PathShadowing.batched_distance(x, y, k, n_splits, cuda=True) (ref path_shadowing.py:97-179): CPU torch
    tensors, distances (B, k) float32 and indices (B, k, 2) int32 -- the reference's own output."""
    import shadowing_amd as sa
    g = load_golden(name)
    ds = rows3(g["dataset"])
    obj = sa.PathShadowing(sa.Identity(g["W"]), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=g["h"]), cache=True)
    x = torch.tensor(g["queries"])[:, None, :]
    d, idx = obj.batched_distance(x, torch.tensor(ds), g["k"], g["n_splits"], cuda=True)
    assert isinstance(d, torch.Tensor) and not d.is_cuda and d.dtype == torch.float32 and tuple(d.shape) == g["d"].shape
    assert not idx.is_cuda and idx.dtype == torch.int32 and tuple(idx.shape) == g["idx"].shape
    assert_matches_reference(d.numpy(), idx.numpy(), g, None, what=name)


@pytest.mark.parametrize("name", SMALL_GOLDENS + ["cfg3_rolling_B128_R256"])
def test_shadow_cuda_identity_matches_the_reference_with_paths(hip_device, oracle_mod, name):
    """PathShadowing(Identity, RelativeMSE).shadow(cuda=True) (ref :181-218): distances, indices AND the gathered
    paths (B, k, C, W + h) against the reference's shadow(cuda=False) output."""
    import shadowing_amd as sa
    g = load_golden(name)
    ds = g["dataset"]
    obj = sa.PathShadowing(sa.Identity(g["W"]), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=g["h"]), cache=True)
    d, paths, idx = obj.shadow(g["queries"] if g["queries"].shape[0] > 1 else g["queries"][0], k=g["k"],
                               n_splits=g["n_splits"], cuda=True)
    assert obj.last_path == "hip"
    assert d.dtype == np.float32 and idx.dtype == np.int32 and paths.dtype == np.float32
    assert paths.shape[:2] == g["d"].shape and paths.shape[2:] == (1, g["W"] + (g["h"] or 0))
    assert_matches_reference(d, idx, g, None, what=name)
    h = g["h"] or 0
    assert np.array_equal(paths[:, :, 0, :], oracle_mod.gather_paths(rows3(ds), idx, g["W"] + h))
    # where the order is determined (no exact ties) the paths ARE the reference's, position by position
    if name not in ("duplicated_paths", "zero_query"):
        same = np.all(idx == g["idx"], axis=-1)
        n = g["paths"].shape[1]
        assert np.array_equal(paths[:, :n][same[:, :n]], g["paths"][same[:, :n]])
        assert same.mean() > 0.99


def test_shadow_cuda_splits_a_batch_of_mixed_amplitudes_into_classes(hip_device, oracle_mod, monkeypatch):
    """PathShadowing looks at the host copy of a batch.  One that will meet the 8-BIT rejection test (32 queries and more,
    W <= 25 -- it puts the queries of a CALL on one quantisation step) and whose amplitudes differ by more than ~3x reaches the
    library as one call per amplitude class of a factor 3, as long as every class keeps 32 queries; otherwise (small classes,
    small batches) the classes are a factor of 64 wide -- what the f16 test's one scale copes with -- and on the f16 test.
    The batch comes back as one, equal to the oracle, either way."""
    import shadowing_amd as sa
    from shadowing_amd import _native
    ds = syn.dataset(2048, 1500, 2800)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(horizon=20))
    seen = []
    real = _native.scan_topk
    monkeypatch.setattr(_native, "scan_topk", lambda *a, **kw: (seen.append((a[1].detach().cpu().numpy().copy(), kw.get("flags", 0))), real(*a, **kw))[1])
    F16 = _native.FLAG_MQ_F16
    two = np.repeat(np.array([1.0, 100.0], np.float32), 40)[:, None]
    cases = {   # name: (B, per-query scale, a zero query?, calls expected (None: 2..4 wide classes), flags of every call)
        "uniform 80":          (80, np.ones((80, 1), np.float32), False, 1, 0),
        "two classes of 40":   (80, two, False, 2, 0),                                  # both keep the 8-bit test
        "... and a zero query": (80, two, True, 3, F16),                                # a class of 39: wide classes on f16 (x1, x100, the zero one)
        "spread 80":           (80, np.geomspace(1.0, 3000.0, 80).astype(np.float32)[:, None], False, None, F16),
        "spread 24":           (24, np.geomspace(1.0, 3000.0, 24).astype(np.float32)[:, None], False, None, 0),
    }
    for name, (B, scale, zero, n_calls, flags) in cases.items():
        x = syn.rolling_queries(B, 20, 2801)
        x = (x / np.abs(x).max(axis=1, keepdims=True) * 0.03 * scale).astype(np.float32)      # every query's largest |sample|: 0.03 x its scale
        if zero:
            x[5] = 0.0
        n0 = len(seen)
        d, paths, idx = obj.shadow(x, k=100, cuda=True)
        calls = seen[n0:]
        shapes = [(c.shape[0], f) for c, f in calls]
        assert all(f & ~_native.FLAG_NO_FUSE == flags for _, f in calls[:1]), (name, shapes)
        if n_calls is not None:
            # the policy's calls come first and partition the batch (whatever follows is the status protocol: an estimate that
            # fell short on this small ensemble -> the checked rerun / the exhaustive pass of those queries)
            assert len(calls) >= n_calls and all(f == flags for _, f in calls[:n_calls]), (name, shapes)
            assert sum(c.shape[0] for c, _ in calls[:n_calls]) == B, (name, shapes)
        else:
            # 2..4 wide classes (plus, at worst, a quiet class's trip through the status protocol)
            first = [c for c, f in calls if f == flags]
            assert 2 <= len(calls) <= 8 and sum(c.shape[0] for c in first[:4]) >= B, (name, shapes)
        if name == "two classes of 40":
            for c, _ in calls[:2]:                                  # within a call: amplitudes within a factor of 3
                a = np.abs(c).max(axis=1)
                assert c.shape[0] == 40 and a.max() <= 3.0 * a.min() * (1 + 1e-5)
        od, oidx = oracle_mod.scan_topk(ds, x, 100, h=20)
        fin = np.isfinite(od).all(axis=1)
        assert fin.sum() >= B - 1
        assert_exact(d[fin], idx[fin], od[fin], oidx[fin], name)
        assert np.array_equal(paths[fin][:, :, 0, :], oracle_mod.gather_paths(rows3(ds), idx[fin], 40))


def test_resident_copy_follows_edits_of_the_ensemble(hip_device):
    """cuda=True: a writeable numpy ensemble is re-read on every call (an in-place edit of one row is seen, as in the
    reference); a torch ensemble stays resident and is re-uploaded when its version counter moves; cache=True keeps
    the HBM copy until refresh()."""
    import shadowing_amd as sa
    q = syn.gbm_log_returns((1, 20), 3100)
    mk = lambda data, **kw: sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), data, sa.PredictionContext(5), **kw)
    ds = syn.dataset(512, 600, 3101)
    for data in (ds.copy(), ds.astype(np.float64), torch.tensor(ds)):
        obj = mk(data)
        d1, _, i1 = obj.shadow(q, k=8, cuda=True)
        data[100, 0, 300:320] = torch.tensor(q[0]) if isinstance(data, torch.Tensor) else q[0]
        d2, _, i2 = obj.shadow(q, k=8, cuda=True)
        assert d1[0, 0] > 0 and d2[0, 0] == 0 and tuple(i2[0, 0]) == (100, 300)
    kept = mk(ds.copy(), cache=True)
    d1, _, _ = kept.shadow(q, k=8, cuda=True)
    kept.dataset[100, 0, 300:320] = q[0]
    assert kept.shadow(q, k=8, cuda=True)[0][0, 0] == d1[0, 0]           # by contract: stale until refresh()
    kept.refresh()
    assert kept.shadow(q, k=8, cuda=True)[0][0, 0] == 0
    assert kept._resident is not None and mk(ds.copy())._resident is None


def test_shadow_cuda_reads_the_status_with_the_results_and_recovers(hip_device, oracle_mod):
    """shadow(cuda=True) enqueues the gather and the copies behind the scan and reads the status WITH the results (one
    synchronisation); a status other than OK -- here: an ensemble of 3000 identical rows, every distance value shared by
    3000 windows, far more ties than the candidate slices hold -- sends the call through the checked path again."""
    import shadowing_amd as sa
    base = syn.dataset(1, 2048, 2100)
    ds = np.ascontiguousarray(np.tile(base, (3000, 1, 1)))
    q = syn.single_query(20, 2101)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=20), cache=True)
    d, paths, idx = obj.shadow(q, k=500, cuda=True)
    assert obj.last_path == "hip"
    od, oidx = oracle_mod.scan_topk(rows3(ds), q[None, :], 500, h=20)
    assert np.array_equal(d.view(np.uint32), od.view(np.uint32)) and np.array_equal(idx, oidx)
    assert np.array_equal(paths[:, :, 0, :], oracle_mod.gather_paths(rows3(ds), idx, 40))
    # and an ordinary ensemble right after, on the same object's workspace
    ds2 = syn.dataset(4096, 2048, 2102)
    obj2 = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds2, sa.PredictionContext(horizon=20), cache=True)
    d2, paths2, idx2 = obj2.shadow(q, k=500, cuda=True)
    od2, oidx2 = oracle_mod.scan_topk(rows3(ds2), q[None, :], 500, h=20)
    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32)) and np.array_equal(idx2, oidx2)
