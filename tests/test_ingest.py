"""Dataset ingestion (shadowing_amd/ingest.py): the reference's batchNNNN.npy files -> one ensemble, whole or
sharded; host logic on CPU, the pinned two-buffer upload on the GPU."""
import numpy as np
import pytest
import torch

from shadowing_amd import ingest, synthetic as syn
from shadowing_amd.distributed import shard_rows


def _write(tmp_path, sizes, C=1, T=64, two_d=False, dtype=np.float32):
    full = syn.dataset(sum(sizes), T, 77) if C == 1 else np.random.default_rng(3).standard_normal((sum(sizes), C, T)).astype(np.float32)
    r = 0
    for i, n in enumerate(sizes):
        part = full[r:r + n]
        np.save(tmp_path / f"batch{i + 1:04}.npy", (part[:, 0, :] if two_d else part).astype(dtype))
        r += n
    (tmp_path / "notes.txt").write_text("ignored")
    np.save(tmp_path / "other.npy", np.zeros(3))
    return full


def test_whole_ensemble_from_batches(tmp_path):
    full = _write(tmp_path, [256, 256, 100])
    info = ingest.describe(tmp_path)
    assert (info["R"], info["C"], info["T"], info["rows"]) == (612, 1, 64, [256, 256, 100])
    ds, off, R = ingest.load_batches(tmp_path, chunk_rows=200)          # chunks straddle file boundaries
    assert off == 0 and R == 612 and ds.dtype == torch.float32 and tuple(ds.shape) == (612, 1, 64)
    assert np.array_equal(ds.numpy(), full)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_tile_the_ensemble(tmp_path, world):
    full = _write(tmp_path, [7, 64, 1, 30], T=32)
    seen = np.zeros(102, bool)
    for rank in range(world):
        ds, off, R = ingest.load_batches(tmp_path, shard=(rank, world), chunk_rows=16)
        lo, hi = shard_rows(102, world, rank)
        assert (off, R, ds.shape[0]) == (lo, 102, hi - lo)
        assert np.array_equal(ds.numpy(), full[lo:hi])
        seen[lo:hi] = True
    assert seen.all()


def test_two_dimensional_and_float64_files(tmp_path):
    full = _write(tmp_path, [10, 5], two_d=True, dtype=np.float64)
    ds, _, _ = ingest.load_batches(tmp_path)
    assert tuple(ds.shape) == (15, 1, 64) and np.array_equal(ds.numpy(), full)


def test_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        ingest.load_batches(tmp_path)
    np.save(tmp_path / "batch0001.npy", np.zeros((4, 1, 16), np.float32))
    np.save(tmp_path / "batch0002.npy", np.zeros((4, 1, 17), np.float32))
    with pytest.raises(ValueError):
        ingest.describe(tmp_path)


def test_path_shadowing_accepts_a_batch_directory(tmp_path, oracle_mod):
    import shadowing_amd as sa
    full = _write(tmp_path, [40, 24], T=200)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), tmp_path, sa.PredictionContext(5))
    q = syn.gbm_log_returns((2, 20), 5)
    d, paths, idx = obj.shadow(q, k=9, cuda=False)
    od, opaths, oidx = oracle_mod.shadow(full, q, 9, 5)
    assert np.array_equal(d.view(np.uint32), od.view(np.uint32)) and np.array_equal(idx, oidx) and np.array_equal(paths, opaths)


@pytest.mark.gpu
def test_pinned_upload_to_the_device(tmp_path, hip_device):
    full = _write(tmp_path, [300, 300, 77], T=128)
    ds, off, R = ingest.load_batches(tmp_path, device=hip_device, chunk_rows=128)
    assert ds.is_cuda and off == 0 and R == 677 and np.array_equal(ds.cpu().numpy(), full)
    part, off2, _ = ingest.load_batches(tmp_path, device=hip_device, shard=(1, 2), chunk_rows=50)
    lo, hi = shard_rows(677, 2, 1)
    assert off2 == lo and np.array_equal(part.cpu().numpy(), full[lo:hi])
