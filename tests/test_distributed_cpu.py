"""The N>1 path on CPU: two processes over gloo run ShardedPathShadowing with the
oracle injected as the per-shard scan and a torch lexsort as the merge; the collective
result must equal the single-process result on the whole ensemble, on every rank."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _oracle_local(ds2d, q, k, h, r_offset):
    import oracle
    d, idx = oracle.scan_topk(ds2d.numpy(), q.numpy(), k, h=h, r_offset=r_offset, nthreads=2)
    return torch.from_numpy(d), torch.from_numpy(idx)


def _torch_merge(d_all, i_all, k):
    d_all, i_all = d_all.numpy(), i_all.numpy()
    B = d_all.shape[0]
    out_d = np.empty((B, k), np.float32)
    out_i = np.empty((B, k, 2), np.int32)
    for b in range(B):
        real = i_all[b, :, 0] >= 0
        dd, ii = d_all[b][real], i_all[b][real]
        o = np.lexsort((ii[:, 1], ii[:, 0], dd))[:k]
        out_d[b], out_i[b] = dd[o], ii[o]
    return torch.from_numpy(out_d), torch.from_numpy(out_i)


def _worker(rank, world, port, R, T, W, h, k, B, tmp):
    sys.path.insert(0, str(REPO))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import shadowing_amd as sa
        from shadowing_amd import synthetic as syn
        from shadowing_amd.distributed import ShardedPathShadowing, shard_rows
        lo, hi = shard_rows(R, world, rank)
        local = syn.dataset_rows(R, T, 5, lo, hi)
        obj = ShardedPathShadowing(sa.Identity(W), sa.RelativeMSE(), local, lo, sa.PredictionContext(h),
                                   local_topk=_oracle_local, merge=_torch_merge)
        q = syn.rolling_queries(B, W, 6)
        d, paths, idx = obj.shadow(q, k)
        np.savez(os.path.join(tmp, f"rank{rank}.npz"), d=d, paths=paths, idx=idx)
        # pipelined form: batch i+1 begins (local scan + start of its all-gather) before batch i is finished
        qs = [torch.tensor(syn.rolling_queries(B, W, 60 + i)) for i in range(3)]
        serial = [obj.scan(qi, k) for qi in qs]
        outs, pend = [], None
        for qi in qs:
            nxt = obj.scan_begin(qi, k)
            if pend is not None:
                outs.append(pend.finish())
            pend = nxt
        outs.append(pend.finish())
        assert pend.finish() is outs[-1]                                   # finishing twice is harmless
        for a, b in zip(serial, outs):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # k beyond the windows of the WHOLE ensemble: the reference's exception type, on every rank alike
        n_all = R * (T - W - h + 1)
        assert obj.n_windows_global() == n_all
        try:
            obj.scan(qs[0], n_all + 1)
            raise AssertionError("k > number of windows must raise")
        except RuntimeError:
            pass
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("R,k", [(37, 50), (5, 300), (1, 50)])   # uneven shards; a shard with fewer than k windows; an EMPTY shard
def test_two_rank_gloo_matches_single_process(tmp_path, oracle_mod, R, k):
    T, W, h, B = 160, 20, 20, 3
    port = 29500 + (os.getpid() % 2000) + (R % 7)
    mp.spawn(_worker, args=(2, port, R, T, W, h, k, B, str(tmp_path)), nprocs=2, join=True)
    from shadowing_amd import synthetic as syn
    ds = syn.dataset(R, T, 5)
    q = syn.rolling_queries(B, W, 6)
    d, paths, idx = oracle_mod.shadow(ds, q, k, h)
    for rank in range(2):
        z = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(z["d"].view(np.uint32), d.view(np.uint32))
        assert np.array_equal(z["idx"], idx)
        assert np.array_equal(z["paths"], paths)


def _worker_embedded(rank, world, port, R, T, h, k, B, tmp):
    sys.path.insert(0, str(REPO))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import shadowing_amd as sa
        from shadowing_amd import synthetic as syn
        from shadowing_amd.distributed import ShardedPathShadowing, shard_rows
        ker = syn.wavelet_bank(3, 64)                                  # (7, 64)
        emb = sa.PathEmbedding(torch.tensor(ker)[:, None, :])

        def local(ds2d, hx, k_, h_, r_offset):
            d, idx = oracle.scan_topk_embedded(ds2d.numpy(), ker, hx.numpy(), k_, h=h_, r_offset=r_offset, nthreads=2)
            return torch.from_numpy(d), torch.from_numpy(idx)

        lo, hi = shard_rows(R, world, rank)
        obj = ShardedPathShadowing(emb, sa.RelativeMSE(), syn.dataset_rows(R, T, 5, lo, hi), lo, sa.PredictionContext(h),
                                   local_topk=local, merge=_torch_merge)
        d, paths, idx = obj.shadow(syn.rolling_queries(B, 64, 6), k)
        np.savez(os.path.join(tmp, f"rank{rank}.npz"), d=d, paths=paths, idx=idx)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_linear_embedding_matches_single_process(tmp_path, oracle_mod):
    """BASELINE configs[4] shape in small: a wavelet filter bank in front of RelativeMSE,
    batched queries, rows sharded over two ranks."""
    R, T, h, k, B = 23, 300, 9, 40, 3
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_embedded, args=(2, port, R, T, h, k, B, str(tmp_path)), nprocs=2, join=True)
    from shadowing_amd import synthetic as syn
    ds = syn.dataset(R, T, 5)
    ker = syn.wavelet_bank(3, 64)
    x = syn.rolling_queries(B, 64, 6)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].numpy()
    d, idx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    paths = oracle_mod.gather_paths(ds, idx, 64 + h)[:, :, None, :]
    for rank in range(2):
        z = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(z["d"].view(np.uint32), d.view(np.uint32))
        assert np.array_equal(z["idx"], idx)
        assert np.array_equal(z["paths"], paths)


def _worker_embedded_dirty(rank, world, port, R, T, h, k, B, tmp):
    sys.path.insert(0, str(REPO))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import shadowing_amd as sa
        from shadowing_amd import synthetic as syn
        from shadowing_amd.distributed import ShardedPathShadowing, shard_rows
        emb = sa.Foveal(alpha=1.4, beta=0.9, max_context=40)
        ker = emb.kernel[:, 0, :].numpy().copy()

        def local(ds2d, hx, k_, h_, r_offset):
            d, idx = oracle.scan_topk_embedded(ds2d.numpy(), ker, hx.numpy(), k_, h=h_, r_offset=r_offset, nthreads=2)
            return torch.from_numpy(d), torch.from_numpy(idx)

        lo, hi = shard_rows(R, world, rank)
        rows = _dirty_rows(R, T)[lo:hi]
        obj = ShardedPathShadowing(emb, sa.RelativeMSE(), rows, lo, sa.PredictionContext(h), local_topk=local, merge=_torch_merge)
        d, paths, idx = obj.shadow(syn.rolling_queries(B, 40, 16), k)
        np.savez(os.path.join(tmp, f"rank{rank}.npz"), d=d, paths=paths, idx=idx)
    finally:
        dist.destroy_process_group()


def _dirty_rows(R, T):
    from shadowing_amd import synthetic as syn
    ds = syn.dataset(R, T, 15).copy()
    ds[R - 3, 0, 100] = np.nan                 # the LAST rank's shard holds the non-finite samples; the first one is clean
    ds[R - 6, 0, 17] = np.inf
    ds[R - 6, 0, 250] = -np.inf
    return ds


def test_two_rank_gloo_dirty_shard_behind_a_linear_embedding(tmp_path, oracle_mod):
    """One rank's shard holds NaN / +-inf samples, the other's is clean, a Foveal embedding in front: every rank returns the
    single-process oracle's answer (the reference's rule: a window is NaN when any tap of the zero-padded kernel meets such a
    sample) -- no rank refuses, none has to know about the other's shard (round 6)."""
    R, T, h, k, B = 21, 320, 7, 60, 2
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker_embedded_dirty, args=(2, port, R, T, h, k, B, str(tmp_path)), nprocs=2, join=True)
    import shadowing_amd as sa
    from shadowing_amd import synthetic as syn
    ds = _dirty_rows(R, T)
    emb = sa.Foveal(alpha=1.4, beta=0.9, max_context=40)
    ker = emb.kernel[:, 0, :].numpy().copy()
    x = syn.rolling_queries(B, 40, 16)
    hx = emb(torch.tensor(x)[:, None, :])[:, 0, :].numpy()
    d, idx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    assert np.isfinite(d).all()
    paths = oracle_mod.gather_paths(ds, idx, 40 + h)[:, :, None, :]
    for rank in range(2):
        z = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(z["d"].view(np.uint32), d.view(np.uint32))
        assert np.array_equal(z["idx"], idx)
        assert np.array_equal(z["paths"], paths, equal_nan=True)


def test_shard_rows_partition():
    from shadowing_amd.distributed import shard_rows
    for R in (1, 7, 8, 262144, 1000003):
        for G in (1, 2, 3, 8):
            blocks = [shard_rows(R, G, g) for g in range(G)]
            assert blocks[0][0] == 0 and blocks[-1][1] == R
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(G - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
