"""NaN / +-inf samples in the ENSEMBLE, the reference's way: its embedding is a conv1d whose kernel is zero-padded by the
horizon (path_embedding.py:48-51), and 0 * NaN = 0 * inf = NaN -- a window is NaN as soon as one sample of
y[r, :, t : t+K+h] is non-finite, and torch.topk(largest=False) ranks it last (path_shadowing.py:165).  Reference goldens
(tests/golden/nan_in_ensemble_*.npz, make_golden.py --nan) through every Identity path -- fused launch, overlap launches,
the 2-3 query launches, the batched scan, the separate launches -- plus the embedded scans, the sharded class and the
C ABI's two preparation entry points against numpy."""
import numpy as np
import pytest
import torch

from _util import NAN_GOLDENS, assert_exact, assert_matches_reference, load_golden, rows3, syn

pytestmark = pytest.mark.gpu


def _dirty(R, T, seed, n_nan=60, n_inf=10, C=1):
    ds = syn.gbm_log_returns((R, C, T), seed)
    g = np.random.default_rng(seed + 1)
    for r, c, t in zip(g.integers(0, R, n_nan), g.integers(0, C, n_nan), g.integers(0, T, n_nan)):
        ds[r, c, t] = np.nan
    for r, c, t in zip(g.integers(0, R, n_inf), g.integers(0, C, n_inf), g.integers(0, T, n_inf)):
        ds[r, c, t] = np.inf if (r + t) % 2 else -np.inf
    return ds


@pytest.mark.parametrize("C,back,fwd", [(1, 0, 0), (1, 20, 0), (3, 7, 5), (2, 300, 11), (1, 0, 9)])
def test_count_and_smear_equal_numpy(hip_device, C, back, fwd):
    from shadowing_amd import _native
    ds = _dirty(130, 257, 900 + C + back, C=C)
    dt = torch.as_tensor(ds).to(hip_device)
    assert _native.count_nonfinite(dt) == int((~np.isfinite(ds)).sum())
    assert _native.count_nonfinite(torch.as_tensor(syn.dataset(33, 1001, 3)).to(hip_device)) == 0
    bad = (~np.isfinite(ds)).any(axis=1)                                   # (R, T): any channel
    want = ds[:, 0, :].copy()
    for r, p in zip(*np.nonzero(bad)):
        want[r, max(0, p - back):p + fwd + 1] = np.nan
    got = _native.smear_nonfinite(dt, back, fwd).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])


@pytest.mark.parametrize("name", NAN_GOLDENS)
def test_shadow_matches_the_reference_with_nonfinite_samples(hip_device, oracle_mod, name):
    """shadow(cuda=True) == the reference's own output (1 query: the fused launch; 3: the overlap launches; 9: the batched
    scan), paths included (gathered from the ensemble itself, NaNs and all)."""
    import shadowing_amd as sa
    g = load_golden(name)
    ds = rows3(g["dataset"])
    obj = sa.PathShadowing(sa.Identity(g["W"]), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(g["h"]))
    d, paths, idx = obj.shadow(g["queries"], k=g["k"], cuda=True)
    assert obj.last_path == "hip"
    all_dist = [oracle_mod.all_distances(ds, q, g["h"]) for q in g["queries"]]
    assert_matches_reference(d, idx, g, all_dist, what=name)
    od, opaths, oidx = oracle_mod.shadow(ds, g["queries"], g["k"], g["h"])
    assert_exact(d, idx, od, oidx, name + " vs oracle")
    assert np.array_equal(paths, opaths, equal_nan=True)
    assert np.isfinite(d).all()                                            # k clean windows exist: no NaN is returned


def test_every_identity_path_with_nonfinite_samples(hip_device, oracle_mod):
    """A larger dirty ensemble (sampled threshold paths, not the tiny-ensemble ones): fused launch, shadow_async (overlap
    launches), 2 and 3 queries, a batch, and the separate launches through the C ABI on the smeared rows -- all equal
    the oracle, whose rule is pinned on the reference's goldens."""
    import shadowing_amd as sa
    from shadowing_amd import _native
    R, T, W, h, k = 4096, 2048, 20, 20, 300
    ds = _dirty(R, T, 7100, n_nan=4000, n_inf=500)
    ds[11, 0, :] = np.nan
    obj = sa.PathShadowing(sa.Identity(W), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(h))
    for B in (1, 2, 3, 9):
        q = syn.rolling_queries(B, W, 7200 + B)
        d, paths, idx = obj.shadow(q, k=k, cuda=True)
        od, opaths, oidx = oracle_mod.shadow(ds, q, k, h)
        assert_exact(d, idx, od, oidx, f"dirty ensemble, {B} queries")
        assert np.array_equal(paths, opaths, equal_nan=True)
    q = syn.single_query(W, 7300)
    d, paths, idx = obj.shadow_async(q, k=k).result()
    od, opaths, oidx = oracle_mod.shadow(ds, q[None, :], k, h)
    assert_exact(d, idx, od, oidx, "dirty ensemble, shadow_async")
    # the C ABI on the prepared rows, separate launches and exhaustive path
    dt = torch.as_tensor(ds).to(hip_device)
    rows = _native.smear_nonfinite(dt, h)
    qd = torch.as_tensor(q[None, :]).to(hip_device)
    for kw in (dict(flags=_native.FLAG_NO_FUSE), dict(exhaustive=True)):
        dd, ii, st = _native.scan_topk(rows, qd, k, h=h, **kw)[:3]
        torch.cuda.synchronize()
        assert int(st.max().item()) == 0
        assert_exact(dd.cpu().numpy(), ii.cpu().numpy(), od, oidx, f"dirty ensemble, C ABI {kw}")
    # windows whose FUTURE holds the non-finite sample are exactly what the smear adds: without it the result differs
    dd, ii, _ = _native.scan_topk(dt[:, 0, :].contiguous(), qd, k, h=h, flags=_native.FLAG_NO_FUSE)[:3]
    torch.cuda.synchronize()
    assert not np.array_equal(ii.cpu().numpy(), oidx)


@pytest.mark.parametrize("W,h", [(64, 20), (126, 0), (250, 7), (30, 5)])
def test_long_windows_with_nonfinite_samples(hip_device, oracle_mod, W, h):
    """The long-window scan (one, two / three queries a pass, the loop of steps for more) and the 26 .. 33-tap batches on a
    dirty ensemble: a NaN or an infinity in a segment poisons the f16 tiles of its rows (kept, never rejected), the exact chains
    give NaN for the windows that hold it, and the smeared horizon follows the reference's rule -- all equal the oracle."""
    import shadowing_amd as sa
    R, T, k = 4096, 2048, 200
    ds = _dirty(R, T, 7400 + W, n_nan=1500, n_inf=300)
    ds[5, 0, :] = np.nan
    obj = sa.PathShadowing(sa.Identity(W), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(h))
    for B in (1, 2, 3, 7):
        q = syn.rolling_queries(B, W, 7500 + B)
        d, paths, idx = obj.shadow(q, k=k, cuda=True)
        assert obj.last_path == "hip"
        od, opaths, oidx = oracle_mod.shadow(ds, q, k, h)
        assert_exact(d, idx, od, oidx, f"dirty ensemble, W={W}, {B} queries")
        assert np.array_equal(paths, opaths, equal_nan=True)


@pytest.mark.parametrize("kind", ["foveal", "wavelet", "foveal_one_window", "imputation"])
def test_embedded_scans_with_nonfinite_samples(hip_device, oracle_mod, kind):
    """A dirty ensemble behind a linear embedding stays on the NATIVE path (round 5; rounds 3-4 handed it to torch ops): the
    reference's conv1d makes a window NaN when ANY tap of its zero-padded kernel meets a non-finite sample (also taps no row
    of the kernel spans: Foveal's head, the horizon).  PathShadowing splits the resident ensemble once (psh_rows_nonfinite):
    rows without such a sample keep the sampled embedded scan, the dirty ones go through the exhaustive dense chains on rows
    with the horizon smeared in, the lists are merged (psh_merge_topk) -- bit for bit the oracle's answer (whose NaN rule is
    pinned on the reference's own outputs: tests/golden/nan_in_ensemble_*.npz), no window whose field holds a non-finite
    sample; the same object on a clean ensemble is back on the plain scan."""
    import shadowing_amd as sa
    R, T, h, k = 1024, 1500, 30, 200
    ctx = sa.PredictionContext(h)
    if kind == "foveal":
        emb = sa.Foveal(alpha=1.3, beta=0.9, max_context=60)
    elif kind == "wavelet":
        emb = sa.PathEmbedding(torch.tensor(syn.wavelet_bank(3, 64))[:, None, :])
    elif kind == "foveal_one_window":
        emb = sa.Foveal(alpha=1.3, beta=0.9, max_context=60)
        T = 60 + h                                                  # one window per row: psh_embed_rows + psh_scan_topk
        R, k = 4096, 300
    else:
        emb = sa.Foveal(alpha=1.3, beta=0.9, max_context=40)        # ImputationContext((l, c, r)): l + r = 40 in-context samples
        ctx = sa.ImputationContext((25, 12, 15))
        h = 0
    ds = _dirty(R, T, 7400, n_nan=300 if T > 200 else 40, n_inf=60 if T > 200 else 10)
    K = emb.kernel.shape[-1]
    x = syn.gbm_log_returns((3, K), 7401)
    hx = emb(torch.tensor(x)[:, None, :])[:, 0, :].numpy()
    ker = (ctx.pad_context(emb.kernel) if kind == "imputation" else emb.kernel)[:, 0, :].numpy().copy()
    obj = sa.PathShadowing(emb, sa.RelativeMSE(), torch.as_tensor(ds), ctx)
    for _ in range(2):                                              # (the second call runs on the cached split)
        d, paths, idx = obj.shadow(x, k=k, cuda=True)
        assert obj.last_path == "hip"
        od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
        assert np.isfinite(od).all() and np.isfinite(d).all()
        assert_exact(d, idx, od, oidx, f"dirty ensemble, {kind}")
        L = ker.shape[1] + h
        for b in range(3):
            for r, t in idx[b][:: max(1, k // 50)]:
                assert np.isfinite(ds[r, 0, t:t + L]).all()
        assert np.array_equal(paths[:, :, 0, :], oracle_mod.gather_paths(ds, idx, paths.shape[-1]))
    clean = syn.dataset(R, T, 7402)
    obj.dataset = torch.as_tensor(clean)
    d, paths, idx = obj.shadow(x, k=k, cuda=True)
    assert obj.last_path == "hip" and not obj._dirty
    od, oidx = oracle_mod.scan_topk_embedded(clean, ker, hx, k, h=h)
    assert_exact(d, idx, od, oidx, f"clean ensemble again, {kind}")


def test_dirty_ensemble_with_fewer_clean_windows_than_k(hip_device, oracle_mod):
    """Nearly every row dirty: the exhaustive leg carries the answer, NaN windows rank last (ref path_shadowing.py:165)."""
    import shadowing_amd as sa
    R, T, h, k = 64, 400, 10, 3000
    ds = syn.dataset(R, T, 7600)
    ds[2:, 0, 200] = np.nan                                          # 62 of 64 rows dirty: 2 x 331 clean-row windows < k
    emb = sa.Foveal(alpha=1.3, beta=0.9, max_context=60)
    x = syn.gbm_log_returns((2, 60), 7601)
    hx = emb(torch.tensor(x)[:, None, :])[:, 0, :].numpy()
    obj = sa.PathShadowing(emb, sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(h))
    d, paths, idx = obj.shadow(x, k=k, cuda=True)
    assert obj.last_path == "hip"
    od, oidx = oracle_mod.scan_topk_embedded(ds, emb.kernel[:, 0, :].numpy(), hx, k, h=h)
    assert_exact(d, idx, od, oidx, "mostly dirty")


def test_sharded_class_with_nonfinite_samples(hip_device, oracle_mod):
    import shadowing_amd as sa
    from shadowing_amd.distributed import ShardedPathShadowing
    R, T, W, h, k = 2048, 1024, 20, 20, 128
    ds = _dirty(R, T, 7500, n_nan=900, n_inf=100)
    obj = ShardedPathShadowing(sa.Identity(W), sa.RelativeMSE(), torch.as_tensor(ds), 5000, sa.PredictionContext(h),
                               device=hip_device, always_exchange=True)
    q = syn.rolling_queries(2, W, 7501)
    d, idx = obj.scan(torch.as_tensor(q), k)
    torch.cuda.synchronize()
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h, r_offset=5000)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, "dirty shard")


@pytest.mark.parametrize("kind", ["foveal", "wavelet"])
def test_sharded_class_serves_a_dirty_shard_behind_a_linear_embedding(hip_device, oracle_mod, kind):
    """Round 6: ShardedPathShadowing no longer refuses NaN / +-inf samples behind a linear embedding -- the rank's clean rows keep
    the embedded scan, its dirty rows (horizon smeared in) the exhaustive dense chains, the two lists are merged before the
    exchange (PathShadowing's split, per shard): the oracle's answer, global row numbers, through the all-gather and the merge."""
    import shadowing_amd as sa
    from shadowing_amd.distributed import ShardedPathShadowing
    R, T, h, k = 1024, 1500, 30, 200
    emb = sa.Foveal(alpha=1.3, beta=0.9, max_context=60) if kind == "foveal" else sa.PathEmbedding(torch.tensor(syn.wavelet_bank(3, 64))[:, None, :])
    ds = _dirty(R, T, 7700, n_nan=300, n_inf=60)
    K = emb.kernel.shape[-1]
    x = syn.gbm_log_returns((3, K), 7701)
    hx = emb(torch.tensor(x)[:, None, :])[:, 0, :].numpy()
    obj = ShardedPathShadowing(emb, sa.RelativeMSE(), torch.as_tensor(ds), 3000, sa.PredictionContext(h), device=hip_device,
                               always_exchange=True)
    for _ in range(2):
        d, idx = obj.scan(torch.as_tensor(x), k)
        torch.cuda.synchronize()
        od, oidx = oracle_mod.scan_topk_embedded(ds, emb.kernel[:, 0, :].numpy(), hx, k, h=h, r_offset=3000)
        assert np.isfinite(od).all()
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"dirty shard behind {kind}")
    dn, paths, idn = obj.shadow(x, k=k)
    assert_exact(dn, idn, od, oidx, "shadow() on the dirty shard")
    assert np.array_equal(paths[:, :, 0, :], oracle_mod.gather_paths(ds, idn - np.array([3000, 0], np.int32), paths.shape[-1]))
    obj.close()
