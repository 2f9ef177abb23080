"""psh_shadow_blocking (include/psh.h, version 3): ONE blocking shadow() of one Identity query as one library call -- the fused
launch reads the query from its kernel arguments, ranks, GATHERS the winners' paths itself and writes distances, indices and
paths straight into the caller's pinned block; its last blocks set completion words the host polls.  Against the oracle
(reference path_shadowing.py:181-218: scan, top-k, path gather), through the C ABI (_native.BlockingShadow) and through
PathShadowing.shadow(cuda=True)."""
import numpy as np
import pytest
import torch

from _util import assert_exact
from shadowing_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _slot(dev, ds3, W, k, h):
    from shadowing_amd import _native
    t = torch.as_tensor(ds3).to(dev)
    rows = t[:, 0, :] if t.shape[1] == 1 else t[:, 0, :].contiguous()
    return _native.BlockingShadow(rows, t, W, k, h, _native.Workspace(dev)), t


def _paths_of(ds3, idx, length):
    return np.stack([ds3[r, :, t:t + length] for r, t in idx[0]])[None]


@pytest.mark.parametrize("R,T,W,h,k,C", [
    (4096, 4096, 20, 20, 1024, 1),
    (6000, 2048, 20, 11, 700, 1),
    (3000, 2051, 20, 20, 300, 1),      # T % 4 != 0: the unaligned instantiation (one value in scratch, test_isa_metadata)
    (5000, 1100, 8, 7, 200, 1),        # run-time window lengths
    (5000, 1100, 33, 0, 200, 1),
    (4096, 2048, 20, 20, 500, 3),      # several channels: the scan reads channel 0, the gather every channel
    (2500, 1031, 17, 5, 150, 2),
    (300, 2048, 20, 20, 50, 1),        # fewer units than the launch has waves
])
def test_blocking_entry_equals_oracle(hip_device, oracle_mod, R, T, W, h, k, C):
    g = np.random.default_rng(R + T)
    ds = syn.dataset(R, T, 7000 + R)
    ds3 = ds if C == 1 else np.concatenate([ds] + [(g.standard_normal((R, 1, T)) * 0.01).astype(np.float32) for _ in range(C - 1)], axis=1)
    slot, _ = _slot(hip_device, ds3, W, k, h)
    raw = torch._C._cuda_getCurrentRawStream(hip_device.index)
    for j in range(3):
        q = syn.gbm_log_returns((W,), 7100 + j)
        st, res = slot.call(raw, q, None)
        assert st == 0
        d, paths, idx = res
        od, oidx = oracle_mod.scan_topk(ds, q[None, :], k, h=h)
        assert_exact(d, idx, od, oidx, f"blocking R={R} T={T} W={W} h={h} k={k} C={C} call {j}")
        assert np.array_equal(paths, _paths_of(ds3, oidx, W + h))
        # ... and with the caller's admission level (the hinted launch: no sample phase): the query's own k-th acc x 1.1
        xn2 = float(q.astype(np.float64) @ q.astype(np.float64))
        st, res = slot.call(raw, q, float(od[0, -1]) ** 2 * xn2 * 1.1)
        if st == 0:                                    # (a level that admits more than the launch's lists hold says RETRY: small ensembles)
            assert_exact(res[0], res[2], od, oidx, "hinted blocking call")
            assert np.array_equal(res[1], _paths_of(ds3, oidx, W + h))
        # a hint far too low: fewer than k windows below it -> status, no results
        st, res = slot.call(raw, q, float(od[0, 0]) ** 2 * xn2 * 0.5)
        assert st != 0 and res is None
    assert slot.last_fused or R < 1000


def test_completion_words_protocol_under_repetition(hip_device, oracle_mod):
    """2000 calls, eight queries taking turns, the result region of the block POISONED before every call: whatever the host
    reads after the completion words is what the launch wrote for THIS call (a word that overtook its data, or a stale block,
    shows as a poisoned or a previous query's value)."""
    R, T, W, h, k = 8192, 2048, 20, 20, 512
    ds = syn.dataset(R, T, 7300)
    slot, _ = _slot(hip_device, ds, W, k, h)
    raw = torch._C._cuda_getCurrentRawStream(hip_device.index)
    qs = [syn.gbm_log_returns((W,), 7301 + j) for j in range(8)]
    want = []
    for q in qs:
        od, oidx = oracle_mod.scan_topk(ds, q[None, :], k, h=h)
        want.append((od, oidx, _paths_of(ds, oidx, W + h)))
    blk = slot.blocks[0]
    for c in range(2000):
        j = (c * 5) % 8
        blk.root[slot.o_d:] = 0xff                      # NaN distances, -1 indices, NaN paths
        st, res = slot.call(raw, qs[j], None)
        assert st == 0 and len(slot.blocks) == 1
        d, paths, idx = res
        od, oidx, opaths = want[j]
        ok = np.array_equal(d.view(np.uint32), od.view(np.uint32)) and np.array_equal(idx, oidx) and np.array_equal(paths, opaths)
        assert ok, f"call {c} (query {j}): the block held something else than this call's results when the completion words were seen"
        del d, paths, idx, res


def test_results_handed_out_stay_the_callers(hip_device, oracle_mod):
    """The arrays shadow() returns VIEW the pinned block they were written to: a block is written again only when the caller
    has dropped every array of it.  Twenty results kept (more than the pool of blocks: the later ones are copies) -- all intact
    after twenty more calls; slices of a result keep their block too."""
    import shadowing_amd as sa
    ds = syn.dataset(8192, 2048, 7400)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(horizon=20))
    kept = []
    for i in range(20):
        q = syn.gbm_log_returns((20,), 7401 + i)
        kept.append((q, obj.shadow(q, k=200, cuda=True)))
    part = obj.shadow(kept[0][0], k=200, cuda=True)[1][0, :3]        # a slice survives its parents
    for i in range(20):
        obj.shadow(syn.gbm_log_returns((20,), 7450 + i), k=200, cuda=True)
    for q, (d, paths, idx) in kept:
        od, opaths, oidx = oracle_mod.shadow(ds, q[None, :], 200, 20)
        assert_exact(d, idx, od, oidx, "kept result")
        assert np.array_equal(paths, opaths)
    od, opaths, oidx = oracle_mod.shadow(ds, kept[0][0][None, :], 200, 20)
    assert np.array_equal(part, opaths[0, :3])
    assert obj.last_path == "hip" and obj._sync_slot[1].last_fused


def test_query_forms_the_reference_accepts(hip_device, oracle_mod):
    """shadow() takes the query as numpy or torch, (W,), (1, W) or (1, 1, W), float32 or float64 (ref :16-31, :202), on the host or
    -- as the reference's `_torch` does -- a CUDA tensor, one that requires grad included; all give the same answer."""
    import shadowing_amd as sa
    ds = syn.dataset(4096, 2048, 7500)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), torch.as_tensor(ds).to(hip_device), sa.PredictionContext(horizon=20))
    q = syn.gbm_log_returns((20,), 7501)
    od, opaths, oidx = oracle_mod.shadow(ds, q[None, :], 128, 20)
    forms = [q, q[None, :], q[None, None, :], q.astype(np.float64), torch.as_tensor(q), torch.as_tensor(q)[None, :],
             torch.as_tensor(q).to(hip_device), torch.as_tensor(q).to(hip_device)[None, None, :],
             torch.as_tensor(q).clone().requires_grad_(True), torch.as_tensor(q).to(hip_device).requires_grad_(True)]
    for rep in range(2):                               # (the second round meets the prepared state of the first)
        for x in forms:
            d, paths, idx = obj.shadow(x, k=128, cuda=True)
            assert obj.last_path == "hip"
            assert_exact(d, idx, od, oidx, f"query form {type(x).__name__} {tuple(x.shape)}")
            assert np.array_equal(paths, opaths)
    # shadow_async with a CUDA query (PreparedShadow.launch staged it with .numpy())
    d, paths, idx = obj.shadow_async(torch.as_tensor(q).to(hip_device), k=128).result()
    assert_exact(d, idx, od, oidx, "shadow_async, CUDA query")
    with pytest.raises(Exception):
        obj.shadow(np.zeros(7, np.float32), k=5, cuda=True)


def test_prepared_state_follows_what_changes(hip_device, oracle_mod):
    """The fast way back into the blocking call checks identities only -- so every way a later call can differ must fail one
    of them: another k, another context object or horizon, an in-place edit of a torch ensemble (version counter), a new
    ensemble object, refresh(), several queries, another device-side workspace."""
    import shadowing_amd as sa
    ds = syn.dataset(4096, 2048, 7600)
    t = torch.as_tensor(ds).to(hip_device)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), t, sa.PredictionContext(horizon=20))
    q = syn.gbm_log_returns((20,), 7601)

    def check(dsn, k, h, what):
        d, paths, idx = obj.shadow(q, k=k, cuda=True)
        od, opaths, oidx = oracle_mod.shadow(dsn, q[None, :], k, h)
        assert_exact(d, idx, od, oidx, what)
        assert np.array_equal(paths, opaths), what

    check(ds, 100, 20, "first")
    check(ds, 100, 20, "fast")
    check(ds, 150, 20, "another k")
    obj.context.horizon = 10
    check(ds, 150, 10, "horizon edited in place")
    obj.context = sa.PredictionContext(horizon=None)
    check(ds, 150, 0, "another context object")
    obj.context = sa.PredictionContext(horizon=20)
    check(ds, 150, 20, "back")
    t[7, 0, :] = 0.0                                   # an in-place torch edit bumps the version counter
    ds2 = ds.copy(); ds2[7, 0, :] = 0.0
    check(ds2, 150, 20, "ensemble edited in place")
    ds3 = syn.dataset(2048, 1024, 7602)
    obj.dataset = torch.as_tensor(ds3).to(hip_device)
    check(ds3, 150, 20, "another ensemble")
    qb = syn.rolling_queries(2, 20, 7603)
    d, _, idx = obj.shadow(qb, k=50, cuda=True)
    od, oidx = oracle_mod.scan_topk(ds3, qb, 50, h=20)
    assert_exact(d, idx, od, oidx, "two queries between blocking calls")
    check(ds3, 150, 20, "after a batch")
    # a read-only numpy ensemble kept resident: made writeable and edited -> the next call re-reads it
    arr = syn.dataset(2048, 1024, 7604)
    arr.setflags(write=False)
    obj2 = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), arr, sa.PredictionContext(horizon=20))
    d, _, idx = obj2.shadow(q, k=64, cuda=True)
    d, _, idx = obj2.shadow(q, k=64, cuda=True)
    od, oidx = oracle_mod.scan_topk(arr, q[None, :], 64, h=20)
    assert_exact(d, idx, od, oidx, "read-only numpy ensemble")
    arr.setflags(write=True)
    arr[3, 0, :] = 0.0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        d, _, idx = obj2.shadow(q, k=64, cuda=True)
    od, oidx = oracle_mod.scan_topk(arr, q[None, :], 64, h=20)
    assert_exact(d, idx, od, oidx, "numpy ensemble made writeable and edited")


def test_large_k_and_long_windows_take_the_other_launches(hip_device, oracle_mod):
    """What the fused launch does not serve (k beyond its lists, a window of more than 33 samples) still goes through the one
    blocking call: psh_scan_topk's launches, the gather launch, the stream's end."""
    for R, T, W, h, k in ((8192, 2048, 20, 20, 5000), (4096, 2048, 64, 10, 300)):
        ds = syn.dataset(R, T, 7700 + W)
        slot, _ = _slot(hip_device, ds, W, k, h)
        raw = torch._C._cuda_getCurrentRawStream(hip_device.index)
        q = syn.gbm_log_returns((W,), 7701)
        st, res = slot.call(raw, q, None)
        from shadowing_amd import _native
        if st == _native.PSH_STATUS_RETRY:              # (the three launches of a long window may say so: protocol of psh_scan_topk)
            continue
        assert st == 0 and not slot.last_fused
        od, oidx = oracle_mod.scan_topk(ds, q[None, :], k, h=h)
        assert_exact(res[0], res[2], od, oidx, f"blocking, W={W} k={k}")
        assert np.array_equal(res[1], _paths_of(ds, oidx, W + h))
