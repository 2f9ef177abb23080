"""The oracle is only worth something once it is pinned: check the CPU restatement
(oracle/psh_oracle.c) against every golden vector produced by the reference itself
(tests/golden/make_golden.py ran PathShadowing.shadow(cuda=False) of the read-only
reference).  Distances bit-exact, indices identical modulo the reference's arbitrary
order among exact ties, gathered paths identical."""
import numpy as np
import pytest

from _util import (BIG_GOLDENS, SMALL_GOLDENS, assert_matches_reference, bits, canonical, load_golden, rows3)


@pytest.mark.parametrize("name", SMALL_GOLDENS + BIG_GOLDENS)
def test_oracle_reproduces_reference(oracle_mod, name):
    g = load_golden(name)
    ds = rows3(g["dataset"])
    d, paths, idx = oracle_mod.shadow(ds, g["queries"], g["k"], g["h"])
    assert d.dtype == np.float32 and idx.dtype == np.int32 and paths.dtype == np.float32
    assert d.shape == g["d"].shape and idx.shape == g["idx"].shape and paths.shape == g["paths"].shape
    small = ds.shape[0] * ds.shape[2] <= 1 << 20
    all_dist = None
    if small:
        all_dist = [oracle_mod.all_distances(ds, q, g["h"] or 0) for q in g["queries"]]
    assert_matches_reference(d, idx, g, all_dist, what=name)
    # the reference's own paths, gathered at the reference's own indices, are what the
    # oracle's gather returns for those indices
    W = g["W"]
    h = g["h"] or 0
    ref_paths = oracle_mod.gather_paths(ds, g["idx"], W + h)[:, :, None, :]
    assert np.array_equal(ref_paths, g["paths"])


@pytest.mark.parametrize("name", SMALL_GOLDENS)
def test_oracle_query_norm_is_torchs(oracle_mod, name):
    g = load_golden(name)
    assert np.array_equal(bits(oracle_mod.qnorm(g["queries"])), bits(g["xn"]))


def test_oracle_query_norm_all_lengths(oracle_mod):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(7)
    for W in range(1, 130):
        x = (rng.standard_normal((8, W)) * 0.0126).astype(np.float32)
        assert np.array_equal(bits(oracle_mod.qnorm(x)), bits(torch.tensor(x).norm(dim=-1).numpy())), W


def test_oracle_is_k_minimal_against_brute_force(oracle_mod):
    """top-k == the k smallest of ALL distances in (d, r, t) order."""
    from shadowing_amd import synthetic as syn
    ds = syn.dataset(40, 200, 31)
    q = syn.gbm_log_returns((3, 20), 32)
    k, h = 57, 9
    d, idx = oracle_mod.scan_topk(ds, q, k, h=h)
    for b in range(3):
        full = oracle_mod.all_distances(ds, q[b], h)
        Tp = full.shape[1]
        order = np.lexsort((np.tile(np.arange(Tp), full.shape[0]), np.repeat(np.arange(full.shape[0]), Tp), full.ravel()))[:k]
        assert np.array_equal(bits(full.ravel()[order]), bits(d[b]))
        assert np.array_equal(np.stack([order // Tp, order % Tp], -1).astype(np.int32), idx[b])


def test_oracle_shard_invariance(oracle_mod):
    from shadowing_amd import synthetic as syn
    ds = syn.dataset(64, 256, 33)
    q = syn.gbm_log_returns((2, 20), 34)
    d, idx = oracle_mod.scan_topk(ds, q, 40, h=20)
    parts = [oracle_mod.scan_topk(ds[lo:hi], q, 40, h=20, r_offset=lo) for lo, hi in ((0, 20), (20, 41), (41, 64))]
    dd = np.concatenate([p[0] for p in parts], 1)
    ii = np.concatenate([p[1] for p in parts], 1)
    dd, ii = canonical(dd, ii)
    assert np.array_equal(bits(dd[:, :40]), bits(d)) and np.array_equal(ii[:, :40], idx)


def test_oracle_thread_count_invariance(oracle_mod):
    from shadowing_amd import synthetic as syn
    ds = syn.dataset(128, 300, 35)
    q = syn.gbm_log_returns((1, 20), 36)
    a = oracle_mod.scan_topk(ds, q, 100, h=5, nthreads=1)
    b = oracle_mod.scan_topk(ds, q, 100, h=5, nthreads=5)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1])
