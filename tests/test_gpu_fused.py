"""The single-query step as ONE launch (psh_fused.hip: bootstrap -> threshold -> scan -> distributed selection, blocks
exchanging data through tagged granules in the workspace header) against the oracle and the reference's goldens, with
the status protocol of include/psh.h: PSH_STATUS_RETRY -> the same call with PSH_FLAG_NO_FUSE."""
import numpy as np
import pytest
import torch

from _util import assert_exact, assert_matches_reference, load_golden, rows3
from shadowing_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def fused_scan(dev, ds, q, k, h, ws=None, **kw):
    """One raw call (no retry): (d, idx, status, info)."""
    from shadowing_amd import _native
    ds_t = ds if isinstance(ds, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(rows3(ds)[:, 0, :])).to(dev)
    q_t = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(q), dtype=np.float32)).to(dev)
    info = {}
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws, info=info, **kw)
    torch.cuda.synchronize(dev)
    return d.cpu().numpy(), idx.cpu().numpy(), st.cpu().numpy(), info


def checked_scan(dev, ds, q, k, h, ws=None):
    from shadowing_amd import _native
    ds_t = torch.as_tensor(np.ascontiguousarray(rows3(ds)[:, 0, :])).to(dev)
    q_t = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(q), dtype=np.float32)).to(dev)
    d, idx = _native.scan_topk_checked(ds_t, q_t, k, h=h, workspace=ws)
    torch.cuda.synchronize(dev)
    return d.cpu().numpy(), idx.cpu().numpy()


@pytest.mark.parametrize("R,T,W,h,k", [
    (4096, 4096, 20, 20, 1024),     # a quarter of the rows sampled
    (6000, 2048, 20, 11, 700),
    (2048, 2048, 20, 0, 200),
    (3000, 2051, 20, 20, 300),      # T % 4 != 0: unaligned rows, ragged last segment
    (5000, 1100, 8, 7, 200),        # run-time window lengths
    (5000, 1100, 17, 7, 200),
    (5000, 1100, 33, 7, 200),
    (40000, 1024, 20, 20, 3000),    # k close to what the front lists are sized for
    (1200, 9000, 20, 20, 64),       # long rows: 9 segments per row
])
def test_fused_launch_equals_oracle(hip_device, oracle_mod, R, T, W, h, k):
    ds = syn.dataset(R, T, 4000 + R)
    q = syn.gbm_log_returns((1, W), 4100 + W)
    d, idx, status, info = fused_scan(hip_device, ds, q, k, h)
    assert info["path"] == 2, "the fused launch must be the path taken for a single query on the sampled path"
    assert status[0] == 0, "ordinary data: the fused launch serves the call itself"
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, f"fused R={R} T={T} W={W} h={h} k={k}")


def test_fused_launch_matches_the_reference_at_configs1_size(hip_device):
    g = load_golden("cfg2_R32768")
    d, idx, status, info = fused_scan(hip_device, g["dataset"], g["queries"], g["k"], g["h"])
    assert info["path"] == 2 and status[0] == 0
    assert_matches_reference(d, idx, g, None, what="cfg2 fused")
    d2, idx2, st2, info2 = fused_scan(hip_device, g["dataset"], g["queries"], g["k"], g["h"], flags=16)   # PSH_FLAG_NO_FUSE
    assert info2["path"] == 0 and st2[0] == 0
    assert_exact(d, idx, d2, idx2, "fused vs separate launches")


def test_one_workspace_many_launches(hip_device, oracle_mod):
    """The header's epoch advances from launch to launch; tags of one launch never satisfy the next.  60 calls back to
    back on one workspace (no synchronisation in between), three different queries in rotation, then every result
    checked."""
    from shadowing_amd import _native
    ds = syn.dataset(8192, 2048, 4200)
    ds_t = torch.as_tensor(ds[:, 0, :].copy()).to(hip_device)
    qs = [syn.gbm_log_returns((1, 20), 4201 + i) for i in range(3)]
    ws = _native.Workspace(hip_device)
    outs = []
    for i in range(60):
        outs.append(_native.scan_topk(ds_t, torch.as_tensor(qs[i % 3]).to(hip_device), 512, h=20, workspace=ws))
    torch.cuda.synchronize()
    want = [oracle_mod.scan_topk(ds, q, 512, h=20) for q in qs]
    for i, (d, idx, st) in enumerate(outs):
        assert int(st[0]) == 0, i
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), *want[i % 3], f"launch {i}")


def test_unarmed_workspace_is_detected_on_the_device(hip_device, oracle_mod):
    """A workspace psh_workspace_init never saw (arbitrary bytes): the fused launch reports PSH_STATUS_RETRY instead of
    trusting a header it did not write; with PSH_FLAG_NO_FUSE the same buffer serves the separate launches."""
    from shadowing_amd import _native
    ds = syn.dataset(4096, 2048, 4300)
    q = syn.gbm_log_returns((1, 20), 4301)

    class Raw(_native.Workspace):
        def arm(self):
            self.buf.view(torch.int32).random_(0, 2 ** 31 - 1)          # garbage instead of psh_workspace_init

    ws = Raw(hip_device)
    d, idx, status, info = fused_scan(hip_device, ds, q, 300, 20, ws=ws)
    assert info["path"] == 2 and status[0] == 2
    d, idx, status, info = fused_scan(hip_device, ds, q, 300, 20, ws=ws, flags=16)
    assert info["path"] == 0 and status[0] == 0
    od, oidx = oracle_mod.scan_topk(ds, q, 300, h=20)
    assert_exact(d, idx, od, oidx, "separate launches on an unarmed workspace")
    _native.Workspace.arm(ws)                                           # now armed: the fused launch serves it
    d, idx, status, info = fused_scan(hip_device, ds, q, 300, 20, ws=ws)
    assert info["path"] == 2 and status[0] == 0
    assert_exact(d, idx, od, oidx, "after psh_workspace_init")


FUSED_KINDS = ["spikes", "tiny_query", "huge_query", "scale_1e-12", "scale_1e+12", "planted", "student_t", "zero_rows",
               "one_loud_row", "f16_overflow_inf", "loud_data", "quiet_data", "constant"]


def _adversarial(kind, R, T, seed):
    from test_gpu_parity import _adversarial as base
    if kind == "loud_data":                 # the data 1e4 x the query: the scale is set by tau, not by the query
        ds, q = base("spikes", R, T, seed)
        return ds * np.float32(1e4), q
    if kind == "quiet_data":                # the data 1e-4 x the query: every acc ~ ||x||^2, the filter resolves nothing
        ds, q = base("student_t", R, T, seed)
        return ds * np.float32(1e-4), q
    if kind == "constant":                  # every window ties
        ds, q = base("spikes", R, T, seed)
        return np.full_like(ds, 0.01), q
    return base(kind, R, T, seed)


@pytest.mark.parametrize("kind", FUSED_KINDS)
def test_fused_launch_with_adversarial_data_through_the_status_protocol(hip_device, oracle_mod, kind):
    """Whatever the magnitudes: either the fused launch returns the exact result, or it says PSH_STATUS_RETRY and the
    separate launches (then, for ties en masse, the exhaustive path) do.  Never a silently wrong row."""
    R, T, h, k = 12000, 2048, 11, 700
    ds, q = _adversarial(kind, R, T, 4400 + FUSED_KINDS.index(kind))
    d, idx, status, info = fused_scan(hip_device, ds, q, k, h)
    assert info["path"] == 2 and status[0] in (0, 2)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    if status[0] == 0:
        assert_exact(d, idx, od, oidx, kind + " (fused)")
    d2, idx2 = checked_scan(hip_device, ds, q, k, h)
    assert_exact(d2, idx2, od, oidx, kind + " (status protocol)")


def test_given_query_norm_is_used(hip_device, oracle_mod):
    from shadowing_amd import _native
    ds = syn.dataset(4096, 2048, 4500)
    q = syn.gbm_log_returns((1, 20), 4501)
    qn = np.array([0.123], np.float32)
    d, idx, st = _native.scan_topk(torch.as_tensor(ds[:, 0, :].copy()).to(hip_device), torch.as_tensor(q).to(hip_device), 100, h=20,
                                   qnorm=torch.as_tensor(qn).to(hip_device))
    torch.cuda.synchronize()
    assert int(st[0]) == 0
    od, oidx = oracle_mod.scan_topk(ds, q, 100, h=20, qn=qn)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, "qnorm given")


def test_fused_ranking_with_duplicated_rows_and_with_row_offsets_beyond_the_packed_key(hip_device, oracle_mod):
    """Phase D ranks by counting 64-bit keys d << 32 | r << tbits | t when (r, t) packs into 32 bits, and by the
    three-word compare otherwise (a shard's row offset near 2^32 / 2^tbits).  Rows that repeat 4 times make every
    distance occur 4 times: (r, t) decides everywhere, in both forms."""
    base = syn.dataset(2048, 2048, 4300)
    ds = np.ascontiguousarray(np.tile(base, (4, 1, 1)))
    q = syn.gbm_log_returns((1, 20), 4301)
    od, oidx = oracle_mod.scan_topk(ds, q, 512, h=20)
    for off in (0, 777, (1 << 21) - 8192 + 5, 1 << 22):   # 2^11 windows per row: rows up to 2^21 pack
        d, idx, status, info = fused_scan(hip_device, ds, q, 512, 20, r_offset=off)
        assert info["path"] == 2 and status[0] == 0
        oi = oidx.copy(); oi[..., 0] += off
        assert_exact(d, idx, od, oi, f"fused ranking, row offset {off}")
