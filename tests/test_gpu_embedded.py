"""Parity of the embedded scan (psh_scan_topk_embedded: Foveal and user kernels in front of
RelativeMSE) on the GPU: against the reference's golden vectors, and bit for bit against the
CPU oracle's restatement of the same arithmetic order."""
import numpy as np
import pytest
import torch

from _util import CROSS_GOLDENS, EMBEDDED_GOLDENS, IMPUTATION_GOLDENS, PREDICT_GOLDENS, predict_case, assert_exact, assert_matches_reference, bits, load_golden, rows3
from shadowing_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def hip_scan_embedded(dev, ds, ker, hx, k, h, **kw):
    from shadowing_amd import _native
    ds_t = torch.as_tensor(np.ascontiguousarray(rows3(ds)[:, 0, :])).to(dev)
    ker_t = torch.as_tensor(np.ascontiguousarray(ker, dtype=np.float32)).to(dev)
    hx_t = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(hx), dtype=np.float32)).to(dev)
    out = _native.scan_topk_embedded(ds_t, ker_t, hx_t, k, h=h, **kw)
    torch.cuda.synchronize(dev)
    return out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy(), out[3:] and out[3]


@pytest.mark.parametrize("name", EMBEDDED_GOLDENS)
def test_hip_embedded_matches_reference_goldens(hip_device, oracle_mod, name):
    g = load_golden(name)
    ds = rows3(g["dataset"])
    h = g["h"] or 0
    d, idx, status, _ = hip_scan_embedded(hip_device, ds, g["kernel"], g["hx"], g["k"], h)
    assert np.all(status == 0)
    np.testing.assert_allclose(d, np.sort(g["d"], axis=1), rtol=1e-5, atol=0)   # the documented bar
    small = ds.shape[0] * ds.shape[2] <= 1 << 17
    all_dist = [oracle_mod.all_distances_embedded(ds, g["kernel"], q, h) for q in g["hx"]] if small else None
    assert_matches_reference(d, idx, g, all_dist, what=name)                     # what actually holds: bit equality
    od, oidx = oracle_mod.scan_topk_embedded(ds, g["kernel"], g["hx"], g["k"], h=h)
    assert_exact(d, idx, od, oidx, name + " vs oracle")
    # dense kernels small enough for it: the same through the matrix-core rejection test (PSH_FLAG_EMBED_MX -- what
    # PathShadowing passes for every stock non-Foveal embedding: the wavelet bank of configs[4], the user kernel)
    from shadowing_amd import _native
    if g["kernel"].shape[0] <= 12 and not name.startswith("foveal"):
        d2, idx2, st2, _ = hip_scan_embedded(hip_device, ds, g["kernel"], g["hx"], g["k"], h, flags=_native.FLAG_EMBED_MX)
        assert np.all(st2 == 0)
        assert_matches_reference(d2, idx2, g, all_dist, what=name + " (matrix cores)")
        assert_exact(d2, idx2, od, oidx, name + " (matrix cores) vs oracle")


def _foveal_kernel(alpha, beta, K):
    dim = int(np.floor(np.log(K) / np.log(alpha)))
    ker = np.zeros((dim, K), np.float32)
    for i in range(dim):
        n = int(alpha ** (i + 1))
        ker[i, K - n:] = np.float32(n ** (-beta))
    return ker


EMB_CASES = [  # R, T, d, K, h, k, B, kind
    (64, 1024, 8, 24, 20, 64, 1, "dense"),
    (300, 1100, 5, 23, 20, 128, 3, "dense"),       # ragged last segment, K % 4 != 0
    (50, 515, 6, 31, 7, 33, 2, "dense"),           # T % 4 != 0: unaligned rows
    (17, 4096, 0, 126, 0, 1000, 2, "foveal"),      # the tutorial's embedding, exhaustive-sized
    (40, 700, 32, 256, 10, 20, 1, "dense"),        # PSH_MAX_W, d*K = 8192: the LDS limit
    (40, 600, 3, 1, 0, 50, 2, "dense"),            # K = 1
    (6, 2100, 0, 64, 20, 1, 1, "foveal"),          # k = 1
    (2048, 512, 0, 40, 20, 777, 7, "foveal"),      # sampled path, 3 accumulator groups (B = 7)
    (1500, 900, 4, 20, 5, 300, 4, "gaps"),         # rows with leading/trailing/interior zero taps, one all-zero row
    (2048, 700, 100, 80, 5, 300, 3, "suffix"),     # suffix rows, d > 64 (one survivor per verification pass)
    (1024, 1500, 20, 250, 9, 500, 5, "suffix"),    # 16 tap blocks, three accumulator groups
    (4096, 600, 9, 33, 0, 2000, 2, "suffix"),      # many survivors per segment
    (1500, 1100, 6, 23, 20, 128, 9, "dense"),      # 7+ queries: the 512-thread instantiation, 10 queries per pass
    (700, 2100, 11, 60, 5, 300, 23, "dense"),      #   three passes (10 + 10 + 3)
    (2048, 700, 30, 80, 5, 300, 13, "suffix"),     #   suffix rows, 6 queries per pass
]


def _case_inputs(R, T, d, K, B, kind, seed):
    rng = np.random.default_rng(seed)
    ds = syn.dataset(R, T, seed)
    if kind == "foveal":
        ker = _foveal_kernel(1.15 if K > 100 else 1.4, 0.9, K)
    else:
        ker = rng.standard_normal((d, K)).astype(np.float32)
        if kind == "gaps":
            ker[0, :9] = 0; ker[1, 11:] = 0; ker[2, 5:13] = 0; ker[3, :] = 0
    if kind == "suffix":
        # the structure the fast path recognises, in its general form: every row one constant on U & [a_i, K) --
        # U with a gap (an ImputationContext's zero taps), negative constants, duplicates, an all-zero row, d > 64
        U = np.ones(K, bool)
        U[K // 3: K // 3 + 7] = False
        U[:3] = False
        ker = np.zeros((d, K), np.float32)
        starts = rng.integers(3, K, d)
        starts[:4] = [K - 1, K - 1, 3, K // 3 + 2]           # a start inside the gap too
        for i in range(d):
            ker[i, starts[i]:] = np.float32(rng.standard_normal() or 1.0)
        ker[:, ~U] = 0
        ker[5, :] = 0
        ker[7] = ker[6]
    x = syn.gbm_log_returns((B, K), seed + 1)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].numpy()
    return ds, ker, hx


@pytest.mark.parametrize("R,T,d,K,h,k,B,kind", EMB_CASES)
def test_hip_embedded_equals_oracle_seeded(hip_device, oracle_mod, R, T, d, K, h, k, B, kind):
    ds, ker, hx = _case_inputs(R, T, d, K, B, kind, 100 + R + K)
    dd, idx, status, _ = hip_scan_embedded(hip_device, ds, ker, hx, k, h)
    assert np.all(status == 0)
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    assert_exact(dd, idx, od, oidx, f"embedded {(R, T, d, K, h, k, B, kind)}")


@pytest.mark.parametrize("R,T,d,K,h,k,B,kind", [(2048, 512, 0, 40, 20, 777, 7, "foveal"), (300, 1100, 5, 23, 20, 128, 3, "dense")])
def test_embedded_exhaustive_path_equals_oracle(hip_device, oracle_mod, R, T, d, K, h, k, B, kind):
    ds, ker, hx = _case_inputs(R, T, d, K, B, kind, 7 + R)
    dd, idx, status, _ = hip_scan_embedded(hip_device, ds, ker, hx, k, h, exhaustive=True)
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    assert_exact(dd, idx, od, oidx, "embedded exhaustive")


@pytest.mark.parametrize("kind,d,K", [("foveal", 0, 126), ("suffix", 40, 60)])
def test_suffix_rows_fast_path_equals_the_dense_chains(hip_device, kind, d, K):
    """Foveal-like kernels take the bound-then-verify pass over running sums; PSH_FLAG_EMBED_DENSE forces the dense
    chains.  Same bits either way -- also with non-finite samples in the ensemble (NaN is kept by the cheap test,
    an infinite segment maximum disarms it)."""
    ds, ker, hx = _case_inputs(2048, 1024, d, K, 3, kind, 31)
    ds = ds.copy()
    ds[5, 0, 100] = np.nan
    ds[700, 0, 513] = np.inf
    ds[701, 0, 17] = -np.inf
    ds[1500, 0, 900:903] = 3.0e38
    fast = hip_scan_embedded(hip_device, ds, ker, hx, 512, 20, profile=True)
    from shadowing_amd import _native
    dense = hip_scan_embedded(hip_device, ds, ker, hx, 512, 20, profile=True, flags=_native.FLAG_EMBED_DENSE)
    assert np.all(fast[2] == 0) and np.all(dense[2] == 0)
    assert_exact(fast[0], fast[1], dense[0], dense[1], "suffix rows vs dense")
    # the bootstrap of the fast path hands over upper bounds (a 1e-4 relative margin): a few more candidates -- when the
    # sampled minima are taken over the same groups of windows (the tap walk: 16 consecutive windows per lane, like the dense
    # chains; the prefix-sum variant of Foveal groups 16 windows 64 apart, another estimate of the same level)
    taps = hip_scan_embedded(hip_device, ds, ker, hx, 512, 20, profile=True, flags=_native.FLAG_EMBED_TAPS)
    assert np.all(taps[2] == 0)
    assert_exact(taps[0], taps[1], dense[0], dense[1], "suffix rows (tap walk) vs dense")
    assert dense[3]["n_candidates"] <= taps[3]["n_candidates"] <= 1.05 * dense[3]["n_candidates"] + 8
    assert 512 <= fast[3]["n_candidates"] <= 1.3 * dense[3]["n_candidates"] + 8


def _interval_kernel(rng, d, K, top, dup=0, empty=False):
    """Rows of one constant on [a_i, top): the structure the prefix-sum scan takes (psh_embed_px.hip) -- `dup` copies of one
    row (merged four to a group), negative constants, an optional all-zero row, zero taps above `top`."""
    ker = np.zeros((d, K), np.float32)
    starts = rng.integers(0, top, d)
    starts[0] = 0
    if d > 1:
        starts[1] = top - 1
    for i in range(d):
        ker[i, starts[i]:top] = np.float32(rng.standard_normal() or 1.0)
    for i in range(dup):
        ker[2 + i] = ker[2]
    if empty:
        ker[d - 1] = 0
    return ker


PX_CASES = [  # R, T, d, K, top, dup, empty, h, k, B : what it exercises
    (2048, 700, 12, 61, 61, 7, True, 5, 300, 3),      # K % 4 == 1 (the last prefix entry has a group of its own), 7 copies of a row, an empty row
    (2048, 701, 9, 33, 30, 0, False, 0, 500, 2),      # T % 4 != 0: unaligned rows; zero taps above the interval
    (1024, 1500, 20, 256, 256, 3, False, 9, 400, 5),  # K = 256: the longest window
    (2048, 640, 70, 48, 48, 12, False, 5, 300, 9),    # 70 rows, 12 of them identical: 61 merged rows; 7+ queries (512 threads, 6 a pass)
    (1500, 900, 1, 20, 20, 0, False, 5, 300, 1),      # one row
]


@pytest.mark.parametrize("R,T,d,K,top,dup,empty,h,k,B", PX_CASES)
def test_prefix_sum_scan_equals_oracle_and_tap_walk(hip_device, oracle_mod, R, T, d, K, top, dup, empty, h, k, B):
    from shadowing_amd import _native
    rng = np.random.default_rng(1000 + d + K)
    ker = _interval_kernel(rng, d, K, top, dup, empty)
    ds = syn.dataset(R, T, 500 + K)
    x = syn.gbm_log_returns((B, K), 501 + K)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].numpy()
    ws = _native.Workspace(hip_device)
    dd, idx, status, prof = hip_scan_embedded(hip_device, ds, ker, hx, k, h, profile=True, workspace=ws)
    plan = _native.embed_plan(ws)
    assert prof["path"] == 0 and plan["one_interval"] and plan["ktop"] == top and plan["d"] == d
    assert plan["merged_rows"] <= d - max(0, dup - (dup + 3) // 4)
    assert np.all(status == 0)
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    assert_exact(dd, idx, od, oidx, "prefix sums vs oracle")
    td, tidx, tstatus, _ = hip_scan_embedded(hip_device, ds, ker, hx, k, h, flags=_native.FLAG_EMBED_TAPS)
    assert np.all(tstatus == 0)
    assert_exact(dd, idx, td, tidx, "prefix sums vs tap walk")


def test_kept_plan_and_changed_kernels(hip_device, oracle_mod):
    """keep_plan: the second call with the same kernel tensor skips the plan launch (PSH_FLAG_EMBED_PLAN_KEEP) -- same
    results; an in-place edit of the tensor, another tensor or a grown workspace are noticed and planned afresh."""
    from shadowing_amd import _native
    rng = np.random.default_rng(5)
    R, T, K, h, k, B = 2048, 700, 61, 5, 300, 2
    ds = syn.dataset(R, T, 91)
    ds_t = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :])).to(hip_device)
    x = syn.gbm_log_returns((B, K), 92)
    ws = _native.Workspace(hip_device)

    def run(ker_np, ker_t, kk=k):
        hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker_np)[:, None, :])[:, :, 0]
        out = _native.scan_topk_embedded(ds_t, ker_t, hx.contiguous().to(hip_device), kk, h=h, workspace=ws, keep_plan=True)
        torch.cuda.synchronize()
        od, oidx = oracle_mod.scan_topk_embedded(ds, ker_np, hx.numpy(), kk, h=h)
        assert np.all(out[2].cpu().numpy() == 0)
        assert_exact(out[0].cpu().numpy(), out[1].cpu().numpy(), od, oidx, "kept plan")

    k1 = _interval_kernel(rng, 12, K, K, 3)
    t1 = torch.tensor(k1).to(hip_device)
    run(k1, t1)
    assert ws.plan_of is not None and ws.plan_of[0] is t1
    run(k1, t1)                                               # kept
    k2 = k1.copy(); k2[:, 20:27] = 0                           # a gap: another structure, same tensor edited in place
    t1.copy_(torch.tensor(k2))
    run(k2, t1)
    assert not _native.embed_plan(ws)["one_interval"]
    k3 = _interval_kernel(rng, 9, K, K - 4)
    t3 = torch.tensor(k3).to(hip_device)
    run(k3, t3)                                               # another tensor
    assert _native.embed_plan(ws)["one_interval"] and _native.embed_plan(ws)["d"] == 9
    run(k3, t3, kk=12000)                                     # a larger k: the workspace grows, the plan with it
    run(k3, t3, kk=12000)


def test_kernels_the_prefix_sum_scan_does_not_take(hip_device, oracle_mod):
    """A gap in the common support (ImputationContext), more than 64 distinct rows, a row that is not one constant: the
    plan says so on the device and the tap walk / the dense chains do the work -- same results."""
    from shadowing_amd import _native
    rng = np.random.default_rng(77)
    R, T, K, h, k, B = 2048, 700, 80, 5, 300, 2
    ds = syn.dataset(R, T, 78)
    x = syn.gbm_log_returns((B, K), 79)
    many = _interval_kernel(rng, 100, K, K)                       # ~70 distinct starts: more than 64 merged rows
    gap = _interval_kernel(rng, 10, K, K); gap[:, 30:37] = 0
    ragged = _interval_kernel(rng, 10, K, K); ragged[4, K - 2] *= 2
    for name, ker in (("many", many), ("gap", gap), ("ragged", ragged)):
        hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].numpy()
        ws = _native.Workspace(hip_device)
        dd, idx, status, prof = hip_scan_embedded(hip_device, ds, ker, hx, k, h, profile=True, workspace=ws)
        plan = _native.embed_plan(ws)
        assert prof["path"] == 0 and not plan["one_interval"], (name, plan)
        assert np.all(status == 0)
        od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
        assert_exact(dd, idx, od, oidx, name)


def test_estimated_threshold_that_falls_short_raises_the_status(hip_device, oracle_mod):
    """The embedded scan admits below an ESTIMATE of the k-th smallest distance taken from the sampled rows.  Here the
    sample lies: near-copies of the query sit in the sampled rows only, so fewer than k windows of the whole ensemble
    are below the estimate.  The selection must say so (status != 0) and the exhaustive path must then return the
    exact answer -- what PathShadowing does with such a status."""
    R, T, K, k, h = 2048, 640, 40, 1000, 0
    ds, ker, hx = _case_inputs(R, T, 0, K, 1, "foveal", 77)
    _, _, _, prof = hip_scan_embedded(hip_device, ds, ker, hx, k, h, profile=True)
    n_s = prof["n_sample_rows"]
    assert prof["path"] == 0 and 0 < n_s < R
    stride = R // n_s
    x = syn.gbm_log_returns((1, K), 78)[0]
    ds = ds.copy()
    rng = np.random.default_rng(5)
    rank2 = (3 * k * n_s + 2 * R - 1) // (2 * R) + 16            # the rank of the estimate (psh_capi.hip)
    assert rank2 < k
    planted = 0
    for i in range(n_s):                                     # one near-copy per 65-sample stretch of every sampled row: each
        r = stride // 2 + i * stride                         # in a group of its own, whether a lane's 16 windows are consecutive
        for t in range(0, T - K - 64, 65):                   # (tap walk, dense chains) or 64 apart (prefix sums)
            if planted < rank2 + 40:
                ds[r, 0, t:t + K] = x * (1 + 1e-3 * rng.standard_normal(K)).astype(np.float32)
                planted += 1
    assert rank2 < planted < k
    dd, idx, status, _ = hip_scan_embedded(hip_device, ds, ker, hx, k, h)
    assert status[0] != 0, "the estimate fell short and nobody noticed"
    dd, idx, status, _ = hip_scan_embedded(hip_device, ds, ker, hx, k, h, exhaustive=True)
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    assert_exact(dd, idx, od, oidx, "exhaustive after a short estimate")


def test_embedded_sampled_path_is_taken(hip_device):
    ds, ker, hx = _case_inputs(2048, 2048, 0, 126, 2, "foveal", 5)
    _, _, status, prof = hip_scan_embedded(hip_device, ds, ker, hx, 1024, 252, profile=True)
    assert np.all(status == 0) and prof["path"] == 0 and prof["n_sample_rows"] > 0
    assert 1024 <= prof["n_candidates"] < 200 * 1024


def test_identity_kernel_through_the_embedded_scan_is_the_plain_scan(hip_device, oracle_mod):
    ds = syn.dataset(200, 1024, 71)
    q = syn.gbm_log_returns((2, 20), 72)
    dd, idx, status, _ = hip_scan_embedded(hip_device, ds, np.eye(20, dtype=np.float32), q, 100, 20)
    od, oidx = oracle_mod.scan_topk(ds, q, 100, h=20)
    assert_exact(dd, idx, od, oidx, "eye(20)")


def test_embedded_argument_errors(hip_device):
    from shadowing_amd import _native
    ds = torch.zeros((8, 600), device=hip_device)
    hx = torch.ones((1, 128), device=hip_device)
    with pytest.raises(_native.NativeLibraryError):          # d * K over the LDS limit -> unsupported
        _native.scan_topk_embedded(ds, torch.ones((128, 256), device=hip_device), hx, 4)
    assert not _native.embedding_supported(128, 256) and _native.embedding_supported(34, 126)
    assert _native.embedding_supported(64, 256) and _native.embedding_supported(39, 252)
    with pytest.raises(_native.NativeLibraryError):          # one window per row: psh_embed_rows + psh_scan_topk instead
        _native.scan_topk_embedded(ds, torch.ones((4, 600), device=hip_device)[:, :256].contiguous()[:, :200].contiguous(),
                                   torch.ones((1, 4), device=hip_device), 4, h=400)
    with pytest.raises(ValueError):                           # k larger than the number of windows
        _native.scan_topk_embedded(ds, torch.ones((4, 500), device=hip_device)[:, :200].contiguous(),
                                   torch.ones((1, 4), device=hip_device), 8 * 401 + 1)


@pytest.mark.parametrize("name", ["user_kernel_d20_one_window_rows", "user_kernel_d9_one_window_rows", "foveal_one_window_rows"])
def test_one_window_rows_behind_a_linear_embedding(hip_device, oracle_mod, name):
    """T == K + h: psh_embed_rows + psh_scan_topk over the R embedded points (rows_kernel: the 8-lane reduce over d
    the reference uses for that layout) -- bit for bit the oracle; the reference itself bit for bit where its conv1d's
    own tap order is the plain chain (the user kernels), 1e-6 with identical indices on the Foveal fixture; and the
    same through PathShadowing.shadow(cuda=True)."""
    from shadowing_amd import _native
    import shadowing_amd as sa
    g = load_golden(name)
    ds = rows3(g["dataset"])
    h = g["h"] or 0
    ds_t = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :])).to(hip_device)
    ker_t = torch.as_tensor(g["kernel"]).to(hip_device)
    points = _native.embed_rows(ds_t, ker_t)
    d, idx, status = _native.scan_topk(points, torch.as_tensor(g["hx"]).to(hip_device), g["k"], h=0)
    torch.cuda.synchronize()
    assert not status.cpu().numpy().any()
    od, oidx = oracle_mod.scan_topk_embedded(ds, g["kernel"], g["hx"], g["k"], h=h)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, name + " vs oracle")
    obj = sa.PathShadowing(sa.PathEmbedding(torch.tensor(g["kernel"])[:, None, :]), sa.RelativeMSE(), g["dataset"],
                           sa.PredictionContext(horizon=g["h"]), cache=True)
    d2, paths, idx2 = obj.shadow(g["queries"], k=g["k"], cuda=True)
    assert obj.last_path == "hip"
    assert_exact(d2, idx2, od, oidx, name + " through PathShadowing")
    if name.startswith("user_kernel"):
        assert_matches_reference(d2, idx2, g, None, what=name)
    else:
        np.testing.assert_allclose(d2, np.sort(g["d"], 1), rtol=1e-6, atol=0)
    assert np.array_equal(paths[:, :, 0, :], oracle_mod.gather_paths(ds, idx2, g["kernel"].shape[1] + h))


def test_kernel_matrix_larger_than_sixteen_tiles_allow(hip_device, oracle_mod):
    """Foveal(1.15, 0.9, 252) -- 39 x 252 taps, 39 KB of LDS: runs the 8-wave instantiation of the embedded scan for
    any batch size (it used to fall back to the generic torch path)."""
    import shadowing_amd as sa
    fov = sa.Foveal(alpha=1.15, beta=0.9, max_context=252)
    assert tuple(fov.kernel.shape) == (39, 1, 252)
    ds = syn.dataset(1024, 2048, 88)
    x = syn.gbm_log_returns((2, 252), 89)
    obj = sa.PathShadowing(fov, sa.RelativeMSE(), ds, sa.PredictionContext(horizon=20), cache=True)
    d, _, idx = obj.shadow(x, k=300, cuda=True)
    assert obj.last_path == "hip"
    hx = fov(torch.tensor(x)[:, None, :])[:, 0, :].numpy()
    od, oidx = oracle_mod.scan_topk_embedded(ds, fov.kernel[:, 0, :].numpy(), hx, 300, h=20)
    assert_exact(d, idx, od, oidx, "Foveal max_context 252")


@pytest.mark.parametrize("R,T,d,K,h,k,B,kind", [
    (2048, 2048, 11, 252, 20, 200, 3, "wavelet"),     # BASELINE configs[4]'s bank: three row groups, nine 32-tap steps
    (4096, 1024, 5, 23, 7, 100, 2, "dense"),          # two row groups (rows 5..7 zero), one step
    (3000, 1100, 12, 64, 3, 300, 5, "dense"),         # d = 12: every tile used; ragged last segment
    (5000, 515, 1, 7, 0, 50, 1, "dense"),             # one row; unaligned rows (T % 4 != 0)
    (4096, 4096, 11, 252, 20, 1024, 16, "wavelet"),   # configs[4]'s query batch
    (1500, 1200, 9, 256, 0, 64, 2, "dense"),          # PSH_MAX_W taps
    (20000, 300, 4, 16, 5, 400, 1, "dense"),          # short rows: one segment each, mostly inadmissible windows
    (600, 1100, 7, 40, 3, 150, 9, "dense"),           # the per-query pass on the matrix cores: 9 queries (a ragged group of 4), d <= 8: one K half
    (300, 2100, 4, 33, 0, 500, 23, "dense"),          #   fewer units than waves: the queries split into groups of their own; one row group
    (20000, 300, 10, 16, 5, 400, 6, "dense"),         #   short rows: inadmissible windows in every segment, 6 queries
    (2048, 1500, 11, 200, 9, 2000, 4, "wavelet"),     #   many survivors per unit (k = 2000 of 2.6 M windows): the staged exact chains, 4 at a time
])
def test_dense_kernel_on_the_matrix_cores_equals_oracle(hip_device, oracle_mod, R, T, d, K, h, k, B, kind):
    """PSH_FLAG_EMBED_MX (embed_mx_kernel: hi/lo f16 banded product as the rejection test, exact dense chains for the
    survivors): bit for bit the oracle, and the same candidates decide as without the flag."""
    from shadowing_amd import _native
    ds = syn.dataset(R, T, 500 + R)
    if kind == "wavelet":
        ker = syn.wavelet_bank((d - 1) // 2, K)
    else:
        ker = (np.random.default_rng(600 + d).standard_normal((d, K)) * 0.3).astype(np.float32)
        ker[:, ::3] *= 1e-3                                   # a wide dynamic range inside the rows: the lo parts matter
    x = syn.gbm_log_returns((B, K), 700 + K)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].numpy()
    dd, idx, status, prof = hip_scan_embedded(hip_device, ds, ker, hx, k, h, profile=True, flags=_native.FLAG_EMBED_MX)
    if status.any():
        bad = np.nonzero(status)[0]
        d2, i2, _, _ = hip_scan_embedded(hip_device, ds, ker, hx[bad], k, h, exhaustive=True)
        dd[bad], idx[bad] = d2, i2
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    assert_exact(dd, idx, od, oidx, f"embed_mx {(R, T, d, K, h, k, B, kind)}")
    # the full scan tests with ONE f16 product (a wider ball around the exact embedding, the same survivors' exact chains
    # decide); PSH_FLAG_EMBED_MX_SPLIT keeps the three split-precision products there too: same results
    sd, sidx, sstatus, _ = hip_scan_embedded(hip_device, ds, ker, hx, k, h, flags=_native.FLAG_EMBED_MX | _native.FLAG_EMBED_MX_SPLIT)
    ok = ~(status.astype(bool) | sstatus.astype(bool))
    assert_exact(dd[ok], idx[ok], sd[ok], sidx[ok], "embed_mx: one product vs the split")
    if prof["path"] == 0 and not status.any():
        assert prof["n_candidates"] >= k              # (its bootstrap samples half segments: another estimate than the plain scan's)


def test_dense_kernel_on_the_matrix_cores_with_awkward_magnitudes(hip_device, oracle_mod):
    """Scales far from one (data 1e-9, kernel 1e+6), a kernel row of zeros, one loud row in the ensemble: the per-segment
    and per-kernel power-of-two scales keep the split exact where it has to be."""
    from shadowing_amd import _native
    R, T, d, K, h, k, B = 3000, 1024, 7, 40, 4, 150, 3
    ds = (syn.dataset(R, T, 801) * np.float32(1e-9)).astype(np.float32)
    ds[77] *= np.float32(1e4)
    ker = (np.random.default_rng(802).standard_normal((d, K)) * 1e6).astype(np.float32)
    ker[3] = 0.0
    x = (syn.gbm_log_returns((B, K), 803) * np.float32(1e-9)).astype(np.float32)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].numpy()
    dd, idx, status, _ = hip_scan_embedded(hip_device, ds, ker, hx, k, h, flags=_native.FLAG_EMBED_MX)
    if status.any():
        bad = np.nonzero(status)[0]
        d2, i2, _, _ = hip_scan_embedded(hip_device, ds, ker, hx[bad], k, h, exhaustive=True)
        dd[bad], idx[bad] = d2, i2
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    assert_exact(dd, idx, od, oidx, "embed_mx, awkward magnitudes")


def test_dense_kernel_on_the_matrix_cores_one_scale_for_a_batch_over_many_decades(hip_device, oracle_mod):
    """The per-query pass puts the whole batch's coordinates into f16 at ONE power-of-two scale (the window energies enter the
    MFMAs as their C operand): 16 queries whose magnitudes span 11 decades, coordinates that are f16-subnormal at that scale
    -- the slack on energies and thresholds keeps the test a rejection test; results are the oracle's."""
    from shadowing_amd import _native
    R, T, d, K, h, k, B = 2048, 1400, 11, 100, 6, 200, 16
    ds = syn.dataset(R, T, 811)
    ker = syn.wavelet_bank((d - 1) // 2, K)
    x = syn.gbm_log_returns((B, K), 812)
    x *= (10.0 ** np.random.default_rng(813).integers(-7, 5, size=B)).astype(np.float32)[:, None]
    for b in (0, 7):                                            # planted near-matches of a quiet and of a loud query
        ds[100 + b, 0, 50:50 + K] = x[b] * np.float32(1.001)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].numpy()
    hx[3, 2::3] *= np.float32(1e-9)                             # coordinates far below their query's largest
    dd, idx, status, _ = hip_scan_embedded(hip_device, ds, ker, hx, k, h, flags=_native.FLAG_EMBED_MX)
    if status.any():
        bad = np.nonzero(status)[0]
        d2, i2, _, _ = hip_scan_embedded(hip_device, ds, ker, hx[bad], k, h, exhaustive=True)
        dd[bad], idx[bad] = d2, i2
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    assert_exact(dd, idx, od, oidx, "embed_mx, one scale for queries over 11 decades")


# ---- through the reference's own API -----------------------------------------------------------------
@pytest.mark.parametrize("name", ["foveal_tutorial_small", "user_kernel_d5_K23", "foveal_ragged_B7"])
def test_path_shadowing_with_linear_embedding_runs_native(hip_device, name):
    """PathShadowing(Foveal / PathEmbedding(kernel), RelativeMSE, ds, PredictionContext).shadow(cuda=True)
    reproduces the reference's shadow(cuda=False) output, through libpsh_hip.so."""
    from shadowing import Foveal, PathEmbedding, PathShadowing, PredictionContext, RelativeMSE
    g = load_golden(name)
    emb = PathEmbedding(torch.tensor(g["kernel"])[:, None, :])
    obj = PathShadowing(emb, RelativeMSE(), g["dataset"], PredictionContext(horizon=g["h"]))
    d, paths, idx = obj.shadow(g["queries"], k=g["k"], n_splits=g["n_splits"], cuda=True)
    assert obj.last_path == "hip"
    assert d.dtype == np.float32 and idx.dtype == np.int32
    assert_matches_reference(d, idx, g, None, what=name)
    K, h = g["kernel"].shape[1], g["h"] or 0
    ds = rows3(g["dataset"])
    for b in range(d.shape[0]):
        for i in (0, d.shape[1] // 2, d.shape[1] - 1):
            r, t = idx[b, i]
            assert np.array_equal(paths[b, i, 0], ds[r, 0, t:t + K + h])
    if name == "foveal_tutorial_small":      # the class itself, not only its kernel
        obj2 = PathShadowing(Foveal(alpha=1.15, beta=0.9, max_context=126), RelativeMSE(), g["dataset"],
                             PredictionContext(horizon=g["h"]))
        d2, _, idx2 = obj2.shadow(g["queries"], k=g["k"], cuda=True)
        assert obj2.last_path == "hip" and np.array_equal(bits(d2), bits(d)) and np.array_equal(idx2, idx)


def test_path_shadowing_keeps_and_renews_its_scanning_kernel(hip_device, oracle_mod):
    """shadow(cuda=True) keeps the scanning kernel on the device between calls (and the library's plan of it); an in-place
    edit of the module's kernel tensor is picked up by the next call."""
    from shadowing import Foveal, PathShadowing, PredictionContext, RelativeMSE
    from shadowing_amd import _native
    ds = syn.dataset(2048, 1024, 61)
    x = syn.gbm_log_returns((2, 40), 62)
    fov = Foveal(alpha=1.4, beta=0.9, max_context=40)
    obj = PathShadowing(fov, RelativeMSE(), ds, PredictionContext(horizon=20))

    def check():
        d, _, idx = obj.shadow(x, k=300, cuda=True)
        assert obj.last_path == "hip"
        ker = fov.kernel[:, 0, :].numpy()
        hx = fov(torch.tensor(x)[:, None, :])[:, 0, :].numpy()
        od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, 300, h=20)
        assert_exact(d, idx, od, oidx, "shadow() with a kept kernel")

    check()
    kept = obj._ker_dev[1]
    check()
    assert obj._ker_dev[1] is kept and obj._workspace.plan_of is not None and obj._workspace.plan_of[0] is kept
    with torch.no_grad():
        fov.kernel[:, :, :7] = 0.25                            # no longer Foveal's structure: rows are not one constant any more
    check()
    assert obj._ker_dev[1] is not kept and not _native.embed_plan(obj._workspace)["one_interval"]


def test_overridden_forward_keeps_the_generic_path(hip_device):
    from shadowing import PathEmbedding, PathShadowing, PredictionContext, RelativeMSE

    class Squared(PathEmbedding):
        def forward(self, x):
            return super().forward(x) ** 2

    ds = syn.dataset(16, 200, 3)
    obj = PathShadowing(Squared(torch.randn(3, 1, 10)), RelativeMSE(), ds, PredictionContext(horizon=2))
    obj.shadow(syn.gbm_log_returns((1, 10), 4), k=5, cuda=True)
    assert obj.last_path == "torch"


def test_sharded_class_with_linear_embedding_over_rccl(hip_device, oracle_mod, tmp_path):
    """BASELINE configs[4] in small: wavelet filter bank (W = 252) + RelativeMSE, batched rolling
    queries, ShardedPathShadowing over a real RCCL process group (one rank: the all-gather
    and the in-place merge still run), logical shards merged like the ranks of a node."""
    import torch.distributed as dist
    import shadowing_amd as sa
    from shadowing_amd import _native
    from shadowing_amd.distributed import ShardedPathShadowing, shard_rows
    R, T, h, k, B = 3000, 1500, 20, 256, 6
    ker = syn.wavelet_bank(5, 252)
    emb = sa.PathEmbedding(torch.tensor(ker)[:, None, :])
    ds = syn.dataset(R, T, 51)
    x = syn.rolling_queries(B, 252, 52)
    hx = emb(torch.tensor(x)[:, None, :])[:, 0, :].numpy()
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx, k, h=h)
    opaths = oracle_mod.gather_paths(ds, oidx, 252 + h)[:, :, None, :]
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1, device_id=hip_device)
    try:
        obj = ShardedPathShadowing(emb, sa.RelativeMSE(), ds, 0, sa.PredictionContext(h), device=hip_device,
                                   always_exchange=True)
        d, paths, idx = obj.shadow(x, k)
        assert_exact(d, idx, od, oidx, "sharded linear embedding")
        assert np.array_equal(paths, opaths)
    finally:
        dist.destroy_process_group()
    # 4 logical shards scanned one after the other on this GPU, merged by the device merge
    parts = []
    for g in range(4):
        lo, hi = shard_rows(R, 4, g)
        o = ShardedPathShadowing(emb, sa.RelativeMSE(), ds[lo:hi], lo, sa.PredictionContext(h), device=hip_device)
        parts.append(o.local_scan(torch.tensor(hx).to(hip_device), k)[:2])
    md, mi = _native.merge_topk(torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1), k)
    assert_exact(md.cpu().numpy(), mi.cpu().numpy(), od, oidx, "4 logical shards")


def test_configs4_wavelet_scan_at_its_per_gpu_size_equals_the_oracle(hip_device, oracle_mod):
    """BASELINE.json configs[4] at ONE GPU's share, exactly what tools/bench_foveal.py --which wavelet times: R = 32768 paths
    x T = 4096, the W = 252 wavelet bank (11 rows), 16 rolling query dates, k = 1024, horizon 20 -- 1.25e8 windows x 16
    queries on the matrix cores (embed_mx_kernel) against the oracle's scan of every window (a few seconds on the box's host
    cores: each window is embedded once for the whole batch), bit for bit."""
    from shadowing_amd import _native
    R, T, K, h, k, B = 32768, 4096, 252, 20, 1024, 16
    ker = syn.wavelet_bank(5, K)
    ds = syn.dataset(R, T, 71)
    x = syn.rolling_queries(B, K, 72)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].contiguous()
    dt = torch.as_tensor(ds[:, 0, :].copy()).to(hip_device)
    kd, hd = torch.tensor(ker).to(hip_device), hx.to(hip_device)
    ws = _native.Workspace(hip_device)
    d, idx, st = _native.scan_topk_embedded(dt, kd, hd, k, h=h, workspace=ws, flags=_native.FLAG_EMBED_MX)[:3]
    torch.cuda.synchronize()
    bad = torch.nonzero(st != 0).flatten()
    if bad.numel():                                          # the estimate fell short for a query: the exact pass, as the host class does
        d2, i2, _ = _native.scan_topk_embedded(dt, kd, hd[bad].contiguous(), k, h=h, workspace=ws, exhaustive=True,
                                               flags=_native.FLAG_EMBED_MX)[:3]
        d[bad], idx[bad] = d2, i2
        torch.cuda.synchronize()
    od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx.numpy(), k, h=h)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, "configs[4] per-GPU size")


def test_wavelet_batches_beyond_one_call_of_the_matrix_core_scan_are_chunked(hip_device, oracle_mod):
    """configs[4] with MANY query dates: 300 / 515 queries are more than embed_mx_kernel's tables take in LDS (256 at d = 11,
    K = 252) -- round 6: psh_scan_topk_embedded serves them as even chunks inside the call (150 + 150, 172 + 172 + 171) instead of
    the vector-ALU scan for all of them (46 -> 10 ms per GPU at 512 queries); per-query status words, the oracle's answer."""
    from shadowing_amd import _native
    R, T, K, h, k = 1024, 2048, 252, 20, 64
    ker = syn.wavelet_bank(5, K)
    ds = syn.dataset(R, T, 81)
    dt = torch.as_tensor(ds[:, 0, :].copy()).to(hip_device)
    kd = torch.tensor(ker).to(hip_device)
    ws = _native.Workspace(hip_device)
    for B in (300, 515):
        x = syn.rolling_queries(B, K, 82 + B)
        hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].contiguous()
        hd = hx.to(hip_device)
        d, idx, st = _native.scan_topk_embedded(dt, kd, hd, k, h=h, workspace=ws, flags=_native.FLAG_EMBED_MX)[:3]
        torch.cuda.synchronize()
        bad = torch.nonzero(st != 0).flatten()
        assert bad.numel() <= B // 4, st.tolist()
        if bad.numel():
            d2, i2, _ = _native.scan_topk_embedded(dt, kd, hd[bad].contiguous(), k, h=h, workspace=ws, exhaustive=True,
                                                   flags=_native.FLAG_EMBED_MX)[:3]
            d[bad], idx[bad] = d2, i2
            torch.cuda.synchronize()
        sel = sorted({0, 1, B // 3, B // 2 - 1, B // 2, B // 2 + 1, B - 2, B - 1, 149, 150, 171, 172})
        sel = [q for q in sel if q < B]
        od, oidx = oracle_mod.scan_topk_embedded(ds, ker, hx.numpy()[sel], k, h=h)
        assert_exact(d.cpu().numpy()[sel], idx.cpu().numpy()[sel], od, oidx, f"wavelet batch of {B}: queries around the chunk boundaries")


@pytest.mark.parametrize("name", IMPUTATION_GOLDENS)
def test_path_shadowing_with_an_imputation_context_runs_native(hip_device, name):
    """PathShadowing(embedding, RelativeMSE, ds, ImputationContext((l, c, r))).shadow(cuda=True) through the
    embedded scan (padded kernel): the reference's shadow(cuda=False) output, paths included."""
    from shadowing import Identity, ImputationContext, PathEmbedding, PathShadowing, RelativeMSE
    g = load_golden(name)
    K = g["kernel"].shape[1]
    emb = Identity(K) if name.startswith("imputation_identity") else PathEmbedding(torch.tensor(g["kernel"])[:, None, :])
    obj = PathShadowing(emb, RelativeMSE(), g["dataset"], ImputationContext(tuple(int(v) for v in g["portion"])))
    d, paths, idx = obj.shadow(g["queries"], k=g["k"], n_splits=g["n_splits"], cuda=True)
    assert obj.last_path == "hip"
    assert_matches_reference(d, idx, g, None, what=name)
    ds = rows3(g["dataset"])
    L = g["kernel_padded"].shape[1]
    for b in range(d.shape[0]):
        for i in range(d.shape[1]):
            r, t = idx[b, i]
            assert np.array_equal(paths[b, i, 0], ds[r, 0, t:t + L])
    # the in-context / out-context split of the gathered paths (what predict_from_paths consumes)
    left, gap, right = (int(v) for v in g["portion"])
    assert obj.context.select_out_context(paths).shape[-1] == gap


@pytest.mark.parametrize("name", CROSS_GOLDENS)
def test_path_shadowing_with_a_cross_channel_context_runs_native(hip_device, name):
    """PathShadowing(embedding, RelativeMSE, ds (R, 1 + oc, T), CrossChannelContext(oc)).shadow(cuda=True): the scan
    runs over channel 0 (the context's zero taps on the other channels contribute nothing), the gathered paths keep
    every channel -- the reference's shadow(cuda=False) output."""
    from shadowing import CrossChannelContext, Foveal, Identity, PathShadowing, RelativeMSE
    g = load_golden(name)
    emb = Identity(20) if name.startswith("crosschannel_identity") else Foveal(alpha=2.0, beta=0.5, max_context=32)
    oc = int(g["out_context_channels"])
    obj = PathShadowing(emb, RelativeMSE(), g["dataset"], CrossChannelContext(oc))
    d, paths, idx = obj.shadow(g["queries"], k=g["k"], n_splits=g["n_splits"], cuda=True)
    assert obj.last_path == "hip"
    assert_matches_reference(d, idx, g, None, what=name)
    W = g["queries"].shape[-1]
    assert paths.shape == g["paths"].shape
    for b in range(d.shape[0]):
        for i in range(d.shape[1]):
            r, t = idx[b, i]
            assert np.array_equal(paths[b, i], g["dataset"][r, :, t:t + W])
    assert obj.context.select_out_context(paths).shape[-2] == oc
    d2, _, idx2 = obj.shadow(g["queries"], k=g["k"], cuda=True)          # second call: resident ensemble + channel-0 copy reused
    assert np.array_equal(bits(d), bits(d2)) and np.array_equal(idx, idx2)
    # ONE query at a time (Identity: the blocking call's prepared slot -- kernels writing the pinned result buffer): the rows
    # of the batch's result, every channel gathered
    # (Foveal: the module's own conv1d embeds one query in another summation order than a batch -- ulps; not compared)
    for b in range(d.shape[0] if name.startswith("crosschannel_identity") else 0):
        d1, p1, i1 = obj.shadow(g["queries"][b], k=g["k"], cuda=True)
        assert obj.last_path == "hip"
        assert np.array_equal(bits(d1[0]), bits(d[b])) and np.array_equal(i1[0], idx[b]) and np.array_equal(p1[0], paths[b])


class _KnownProba:
    """An injected DiscreteProba whose arithmetic is known: weights 1/(1 + rank) normalised, avg = sum w x,
    std = max |x - avg| (deliberately NOT a weighted variance: nothing on the device may assume one)."""

    def __init__(self, distances):
        self.n = distances.shape[1]
        w = 1.0 / (1.0 + np.arange(self.n, dtype=np.float64))
        self.w = w / w.sum()

    def avg(self, x, axis=1):
        assert isinstance(x, np.ndarray) and axis == 1
        return np.tensordot(self.w, np.moveaxis(np.asarray(x, np.float64), 1, 0), axes=1)

    def std(self, x, axis=1):
        return np.abs(np.asarray(x, np.float64) - np.expand_dims(self.avg(x), 1)).max(axis=1)


@pytest.mark.parametrize("proba,eta", [("softmax", 0.1), ("uniform", None), ("known", None)])
def test_predict_keeps_the_paths_on_the_device_when_asked_to(hip_device, monkeypatch, proba, eta):
    """predict(cuda=True, device_predict=True): scan, gather and `to_predict` on the GPU, the averaging by the
    installed DiscreteProba's own avg / std on the (B, k, ...) statistic -- here also an injected class with known
    arithmetic.  Same numbers as shadow() + predict_from_paths() on the host.  Without the opt-in (and without the
    `accepts_torch` marker) a callable is handed numpy arrays, as in the reference."""
    import shadowing
    from shadowing import Foveal, PathShadowing, PredictionContext, RelativeMSE, realized_variance
    ds = syn.dataset(1024, 900, 61)
    x = syn.gbm_log_returns((5, 40), 62)
    obj = PathShadowing(Foveal(alpha=1.4, beta=0.9, max_context=40), RelativeMSE(), ds, PredictionContext(horizon=30))
    if proba == "known":
        monkeypatch.setattr(PathShadowing, "init_averaging_proba", staticmethod(lambda name, d, eta: _KnownProba(d)))
    Ts = [2, 7, 30]
    kinds = []

    def to_predict(p):
        kinds.append(type(p))
        return realized_variance(p, Ts=Ts, vol=False)[:, :, 0, :]

    m, s = obj.predict(x, k=256, to_predict=to_predict, eta=eta, proba_name=proba, n_context_splits=2, cuda=True,
                       device_predict=True)
    assert obj.last_path == "hip" and kinds and all(t is torch.Tensor for t in kinds)
    d, paths, _ = obj.shadow(x, k=256, cuda=True)
    m0, s0 = obj.predict_from_paths(d, paths, lambda p: np.asarray(to_predict(p)), proba, eta)
    np.testing.assert_allclose(m, m0, rtol=2e-6)                         # float32 statistic on either side
    np.testing.assert_allclose(s, s0, rtol=2e-5, atol=1e-12)
    # no opt-in: numpy in, as the reference does -- a torch-incompatible callable works, and gets no tensor
    kinds.clear()
    m1, s1 = obj.predict(x, k=256, to_predict=to_predict, eta=eta, proba_name=proba, cuda=True)
    assert kinds and all(t is np.ndarray for t in kinds)
    np.testing.assert_allclose(m1, m0, rtol=1e-12)
    # the marker on shadowing.realized_variance opts in by itself
    assert getattr(realized_variance, "accepts_torch", False)
    # and a callable that does not return a tensor under device_predict is an error, not a silent fallback
    with pytest.raises(TypeError):
        obj.predict(x, k=16, to_predict=lambda p: 1.0, cuda=True, device_predict=True)


@pytest.mark.parametrize("name", PREDICT_GOLDENS)
@pytest.mark.parametrize("device_predict", [False, True])
def test_predict_cuda_matches_the_reference_predict(hip_device, monkeypatch, name, device_predict):
    """predict(cuda=True[, device_predict=True]) against the REFERENCE's own predict() (PS:256-301) run with the
    known-arithmetic averaging classes of tests/_known_proba.py (make_golden.py --predict): Identity (fused / batched
    scan) and Foveal (prefix-sum scan).  Host statistic: the same float32 paths -> bit-equal; device statistic: a float32
    mean of squares reduced by torch on the GPU instead of numpy on the host -> 2e-6."""
    import shadowing_amd as sa
    obj, g = predict_case(name, sa, monkeypatch)
    Ts = [int(t) for t in g["Ts"]]
    m, s = obj.predict(g["queries"], int(g["k"]), lambda f: sa.realized_variance(f, Ts, vol=True), eta=g["eta"],
                       proba_name=str(g["proba_name"]), n_dataset_splits=int(g["n_dataset_splits"]),
                       n_context_splits=int(g["n_context_splits"]), cuda=True, device_predict=device_predict)
    assert obj.last_path == "hip"
    assert m.shape == g["mean"].shape and s.shape == g["std"].shape
    if device_predict:
        np.testing.assert_allclose(m, g["mean"], rtol=2e-6)
        np.testing.assert_allclose(s, g["std"], rtol=2e-5, atol=1e-12)
    else:
        assert np.array_equal(m, g["mean"]) and np.array_equal(s, g["std"])


def test_reference_test_cell_1_forward_topk_prefix_consistency_on_the_device(hip_device):
    """testing.ipynb:43-53 at its own sizes, device tensors: top-32 (32 splits) == the first 32 of top-64 (64 splits),
    torch.equal on distances and indices -- through the scan kernels (rows_kernel)."""
    import shadowing_amd as sa
    torch.manual_seed(0)
    dist = sa.RelativeMSE()
    x, y = torch.randn(8, 34).to(hip_device), torch.randn(128, 512, 34).to(hip_device)
    d32, i32 = dist.forward_topk(x, y, 32, 32)
    d64, i64 = dist.forward_topk(x, y, 64, 64)
    assert torch.equal(d32, d64[:, :32]) and torch.equal(i32, i64[:, :32])
    assert i32.dtype == torch.int64 and i32.shape == (8, 32, 2) and d32.is_cuda
    for b in range(8):                                       # the indices point at the distances they claim
        r, t = i32[b, :, 0], i32[b, :, 1]
        again = dist(x[b][None, :].cpu(), y[r, t].cpu())
        np.testing.assert_allclose(again.numpy(), d32[b].cpu().numpy(), rtol=1e-6)


def test_reference_test_cell_2_shadow_self_consistency_on_the_device(hip_device):
    """testing.ipynb:62-78 at its own sizes with cuda=True: Foveal(1.15, 0.9, 126), x_context (8,1,126), dataset
    (32,1,4096), horizon 252, k = 1024; re-embedding the returned paths' in-context part reproduces the returned
    distances (the reference asserts rtol 1e-2; here 1e-5), rows ascending."""
    import shadowing_amd as sa
    torch.manual_seed(1)
    emb = sa.Foveal(alpha=1.15, beta=0.9, max_context=126)
    ctx = sa.PredictionContext(horizon=252)
    ds = torch.randn(32, 1, 4096).numpy()
    x = torch.randn(8, 1, 126).numpy()
    obj = sa.PathShadowing(emb, sa.RelativeMSE(), ds, ctx, cache=True)
    d, paths, idx = obj.shadow(x, k=1024, cuda=True)
    assert obj.last_path == "hip" and paths.shape == (8, 1024, 1, 378) and idx.shape == (8, 1024, 2)
    hx = emb(torch.tensor(x))[:, 0, :]
    hp = emb(torch.tensor(ctx.select_in_context(paths)).reshape(-1, 1, 126))[:, 0, :].reshape(8, 1024, -1)
    again = sa.RelativeMSE()(hx[:, None, :], hp).numpy()
    np.testing.assert_allclose(again, d, rtol=1e-5)
    assert np.all(np.diff(d, axis=1) >= 0)
    d_host, _, idx_host = obj.shadow(x, k=1024, cuda=False)   # and the reference's formulation on the host agrees
    np.testing.assert_allclose(d, d_host, rtol=1e-5)
