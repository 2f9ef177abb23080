"""Helpers shared by the tests: golden-vector loading and tie-aware comparison."""
from __future__ import annotations

import json
from functools import lru_cache
from pathlib import Path

import numpy as np

from shadowing_amd import synthetic as syn

GOLDEN = Path(__file__).resolve().parent / "golden"
SMALL_GOLDENS = ["cfg1_h20", "cfg1_hNone", "multiquery_splits", "remainder_split_W12", "oddW33_h11",
                 "W7_h0", "duplicated_paths", "zero_query", "self_match", "single_window_rows",
                 "k1_2d_dataset"]
BIG_GOLDENS = ["cfg2_R32768", "cfg3_rolling_R2048"]
# ImputationContext((l, c, r)): the gap's zero taps are part of the scanning kernel ("kernel_padded")
IMPUTATION_GOLDENS = ["imputation_identity_8_5_12", "imputation_user_kernel_6_9_7"]
# CrossChannelContext(oc): ensemble of 1 + oc channels, the scan reads channel 0, the gathered paths keep all
CROSS_GOLDENS = ["crosschannel_identity_C2", "crosschannel_foveal_C3"]
# linear embeddings (Foveal / user kernels) in front of RelativeMSE: psh_scan_topk_embedded
EMBEDDED_GOLDENS = ["foveal_tutorial_small", "foveal_a2_hNone", "user_kernel_d5_K23", "foveal_ragged_B7",
                    "foveal_tutorial_R1024", "wavelet_W252_rolling"]


@lru_cache(maxsize=4)
def _regen(expr: str) -> np.ndarray:
    return eval(expr, {"__builtins__": {}}, {"dataset": syn.dataset, "single_query": syn.single_query,
                                             "rolling_queries": syn.rolling_queries,
                                             "gbm_log_returns": syn.gbm_log_returns})


def load_golden(name: str) -> dict:
    """Fixture dict; large datasets are regenerated from their seed and checked
    against the stored SHA-256 before anything is compared with them."""
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["meta"] = json.loads(str(g["meta"]))
    g["h"] = None if int(g["h"]) < 0 else int(g["h"])
    for key in ("W", "k", "n_splits"):
        if key in g:
            g[key] = int(g[key])
    if "dataset" not in g:
        ds = _regen(g["meta"]["gen"])
        assert syn.sha256(ds) == str(g["dataset_sha256"]), (
            f"{name}: regenerated dataset differs from the one the reference saw "
            f"(numpy {np.__version__} vs {g['meta']['numpy']})")
        g["dataset"] = ds
    return g


def rows3(ds: np.ndarray) -> np.ndarray:
    ds = np.asarray(ds, dtype=np.float32)
    return ds[:, None, :] if ds.ndim == 2 else ds


def canonical(d: np.ndarray, idx: np.ndarray):
    """Rows reordered by (d, r, t) ascending (NaN last)."""
    d2, i2 = d.copy(), idx.copy()
    for b in range(d.shape[0]):
        o = np.lexsort((idx[b, :, 1], idx[b, :, 0], d[b]))
        d2[b], i2[b] = d[b, o], idx[b, o]
    return d2, i2


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_exact(d, idx, d_ref, idx_ref, what=""):
    """Bit-exact distances and identical indices, position by position."""
    assert d.shape == d_ref.shape and idx.shape == idx_ref.shape, what
    assert np.array_equal(bits(d), bits(d_ref)), f"{what}: distances differ in {np.sum(bits(d) != bits(d_ref))} places"
    assert np.array_equal(idx, idx_ref), f"{what}: indices differ in {np.sum(np.any(idx != idx_ref, -1))} rows"


def assert_matches_reference(d, idx, g: dict, all_dist=None, what=""):
    """Compare a canonical-order result with the reference's raw output modulo the
    reference's arbitrary order inside runs of exactly equal distances:
      * the sorted distance vectors are bit-identical;
      * inside every run of equal d that does not reach the k-th position the index
        SETS are identical;
      * in a run that reaches the k-th position (ties at the boundary: the reference
        keeps an arbitrary subset) every returned index must really have that
        distance (checked against `all_dist` (R, T') when given).
    """
    d_ref, i_ref = canonical(g["d"], g["idx"])
    d_c, i_c = canonical(d, idx)
    assert np.array_equal(bits(d_c), bits(d_ref)), f"{what}: sorted distances differ"
    k = d.shape[1]
    for b in range(d.shape[0]):
        s = 0
        while s < k:
            e = s + 1
            while e < k and bits(d_c[b, e]) == bits(d_c[b, s]):
                e += 1
            mine = {tuple(v) for v in i_c[b, s:e]}
            theirs = {tuple(v) for v in i_ref[b, s:e]}
            if e < k:
                assert mine == theirs, f"{what}: query {b}, run [{s},{e}) index sets differ"
            elif mine != theirs:
                assert len(mine) == e - s, f"{what}: duplicate indices in the boundary run"
                if all_dist is not None:
                    for (r, t) in mine | theirs:
                        assert bits(np.float32(all_dist[b][r, t])) == bits(d_c[b, s]), \
                            f"{what}: boundary-run index ({r},{t}) does not have the tied distance"
            s = e
