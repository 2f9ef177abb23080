"""The single-query step as THREE overlap-friendly launches (psh_stream.hip, PSH_FLAG_OVERLAP: sample + admission level in
one-wave blocks, the barrier-free scan, the ranking) against the oracle and the reference's goldens -- alone, and the way
it is meant to be used: independent queries in flight on several streams, every result checked.  Status protocol of
include/psh.h: PSH_STATUS_RETRY -> the same call with PSH_FLAG_NO_FUSE."""
import numpy as np
import pytest
import torch

from _util import assert_exact, assert_matches_reference, load_golden
from shadowing_amd import synthetic as syn
from test_gpu_fused import FUSED_KINDS, _adversarial, checked_scan, fused_scan

pytestmark = pytest.mark.gpu
OVERLAP = 2048        # PSH_FLAG_OVERLAP


@pytest.mark.parametrize("R,T,W,h,k", [
    (4096, 4096, 20, 20, 1024),
    (6000, 2048, 20, 11, 700),
    (2048, 2048, 20, 0, 200),
    (3000, 2051, 20, 20, 300),      # T % 4 != 0: unaligned rows, ragged last segment
    (5000, 1100, 8, 7, 200),        # run-time window lengths
    (5000, 1100, 17, 7, 200),
    (5000, 1100, 33, 7, 200),
    (40000, 1024, 20, 20, 3000),
    (1200, 9000, 20, 20, 64),       # long rows: 9 segments per row
    (300, 2048, 20, 20, 50),        # fewer units than the scan has waves
])
def test_overlap_launches_equal_oracle(hip_device, oracle_mod, R, T, W, h, k):
    ds = syn.dataset(R, T, 5000 + R)
    q = syn.gbm_log_returns((1, W), 5100 + W)
    d, idx, status, info = fused_scan(hip_device, ds, q, k, h, flags=OVERLAP)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    if info["path"] != 3:                 # too small for the sampled path / the sample: the library's other paths serve it
        assert status[0] == 0
    else:
        assert status[0] == 0, "ordinary data: the three launches serve the call themselves"
    assert_exact(d, idx, od, oidx, f"overlap R={R} T={T} W={W} h={h} k={k}")


def test_overlap_path_is_taken_at_configs1_size_and_matches_the_reference(hip_device):
    g = load_golden("cfg2_R32768")
    d, idx, status, info = fused_scan(hip_device, g["dataset"], g["queries"], g["k"], g["h"], flags=OVERLAP)
    assert info["path"] == 3 and status[0] == 0
    assert_matches_reference(d, idx, g, None, what="cfg2 overlap launches")
    d2, idx2, st2, info2 = fused_scan(hip_device, g["dataset"], g["queries"], g["k"], g["h"])
    assert info2["path"] == 2 and st2[0] == 0
    assert_exact(d, idx, d2, idx2, "overlap launches vs the fused launch")


@pytest.mark.parametrize("n_streams", [2, 3])
def test_independent_queries_on_several_streams(hip_device, oracle_mod, n_streams):
    """What the mode is for: 48 calls with five different queries in rotation, issued round-robin on 2 / 3 streams (one
    workspace per stream, no synchronisation in between): launches of different steps share the chip.  Every result checked."""
    from shadowing_amd import _native
    ds = syn.dataset(16384, 2048, 5200)
    ds_t = torch.as_tensor(ds[:, 0, :].copy()).to(hip_device)
    qs = [syn.gbm_log_returns((1, 20), 5201 + i) for i in range(5)]
    q_t = [torch.as_tensor(q).to(hip_device) for q in qs]
    streams = [torch.cuda.Stream(hip_device) for _ in range(n_streams)]
    wss = [_native.Workspace(hip_device) for _ in range(n_streams)]
    torch.cuda.synchronize()
    outs = []
    for i in range(48):
        with torch.cuda.stream(streams[i % n_streams]):
            outs.append(_native.scan_topk(ds_t, q_t[i % 5], 512, h=20, workspace=wss[i % n_streams], flags=OVERLAP))
    torch.cuda.synchronize()
    want = [oracle_mod.scan_topk(ds, q, 512, h=20) for q in qs]
    for i, (d, idx, st) in enumerate(outs):
        assert int(st[0]) == 0, i
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), *want[i % 5], f"step {i}")


def test_overlap_steps_beside_a_fused_launch_and_a_batched_scan(hip_device, oracle_mod):
    """Nothing in the overlap launches needs the chip to itself: a fused launch (which DOES) and a batched scan run on other
    streams at the same time; the overlap steps stay exact, the fused launch either serves its call or says RETRY."""
    from shadowing_amd import _native
    ds = syn.dataset(16384, 2048, 5300)
    ds_t = torch.as_tensor(ds[:, 0, :].copy()).to(hip_device)
    q1 = syn.gbm_log_returns((1, 20), 5301)
    qb = syn.rolling_queries(24, 20, 5302)
    s = [torch.cuda.Stream(hip_device) for _ in range(3)]
    ws = [_native.Workspace(hip_device) for _ in range(3)]
    torch.cuda.synchronize()
    got = []
    for rep in range(6):
        with torch.cuda.stream(s[0]):
            got.append(("overlap", _native.scan_topk(ds_t, torch.as_tensor(q1).to(hip_device), 300, h=20, workspace=ws[0], flags=OVERLAP)))
        with torch.cuda.stream(s[1]):
            got.append(("fused", _native.scan_topk(ds_t, torch.as_tensor(q1).to(hip_device), 300, h=20, workspace=ws[1])))
        with torch.cuda.stream(s[2]):
            got.append(("batch", _native.scan_topk(ds_t, torch.as_tensor(qb).to(hip_device), 300, h=20, workspace=ws[2])))
    torch.cuda.synchronize()
    w1 = oracle_mod.scan_topk(ds, q1, 300, h=20)
    wb = oracle_mod.scan_topk(ds, qb, 300, h=20)
    for kind, (d, idx, st) in got:
        st = st.cpu().numpy()
        if kind == "fused" and st[0] == 2:
            continue                                    # the fused launch was not co-resident: it says so (status protocol)
        assert (st == 0).all(), kind
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), *(wb if kind == "batch" else w1), kind)


def test_unarmed_workspace_is_detected(hip_device, oracle_mod):
    from shadowing_amd import _native
    ds = syn.dataset(4096, 2048, 5400)
    q = syn.gbm_log_returns((1, 20), 5401)

    class Raw(_native.Workspace):
        def arm(self):
            self.buf.view(torch.int32).random_(0, 2 ** 31 - 1)          # garbage instead of psh_workspace_init

    ws = Raw(hip_device)
    d, idx, status, info = fused_scan(hip_device, ds, q, 300, 20, ws=ws, flags=OVERLAP)
    assert info["path"] == 3 and status[0] == 2
    _native.Workspace.arm(ws)
    d, idx, status, info = fused_scan(hip_device, ds, q, 300, 20, ws=ws, flags=OVERLAP)
    assert info["path"] == 3 and status[0] == 0
    od, oidx = oracle_mod.scan_topk(ds, q, 300, h=20)
    assert_exact(d, idx, od, oidx, "after psh_workspace_init")


@pytest.mark.parametrize("kind", FUSED_KINDS)
def test_overlap_launches_with_adversarial_data_through_the_status_protocol(hip_device, oracle_mod, kind):
    """Whatever the magnitudes: either the three launches return the exact result, or they say PSH_STATUS_RETRY and the
    separate launches (then, for ties en masse, the exhaustive path) do.  Never a silently wrong row."""
    from shadowing_amd import _native
    R, T, h, k = 12000, 2048, 11, 700
    ds, q = _adversarial(kind, R, T, 4400 + FUSED_KINDS.index(kind))
    d, idx, status, info = fused_scan(hip_device, ds, q, k, h, flags=OVERLAP)
    assert info["path"] == 3 and status[0] in (0, 2)
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    if status[0] == 0:
        assert_exact(d, idx, od, oidx, kind + " (overlap launches)")
    ds_t = torch.as_tensor(np.ascontiguousarray(ds if ds.ndim == 2 else ds[:, 0, :])).to(hip_device)
    d2, idx2 = _native.scan_topk_checked(ds_t, torch.as_tensor(np.atleast_2d(q)).to(hip_device), k, h=h, flags=OVERLAP)
    torch.cuda.synchronize()
    assert_exact(d2.cpu().numpy(), idx2.cpu().numpy(), od, oidx, kind + " (status protocol)")


def test_given_query_norm_and_row_offsets(hip_device, oracle_mod):
    """qnorm handed in; duplicated rows (every distance 4 times: (r, t) decides) with row offsets on both sides of what
    packs into the ranking's 64-bit keys."""
    from shadowing_amd import _native
    ds = syn.dataset(4096, 2048, 5500)
    q = syn.gbm_log_returns((1, 20), 5501)
    qn = np.array([0.123], np.float32)
    d, idx, st = _native.scan_topk(torch.as_tensor(ds[:, 0, :].copy()).to(hip_device), torch.as_tensor(q).to(hip_device), 100, h=20,
                                   qnorm=torch.as_tensor(qn).to(hip_device), flags=OVERLAP)
    torch.cuda.synchronize()
    assert int(st[0]) == 0
    od, oidx = oracle_mod.scan_topk(ds, q, 100, h=20, qn=qn)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, "qnorm given")
    base = syn.dataset(2048, 2048, 5502)
    ds4 = np.ascontiguousarray(np.tile(base, (4, 1, 1)))
    od, oidx = oracle_mod.scan_topk(ds4, q, 512, h=20)
    for off in (0, 777, (1 << 21) - 8192 + 5, 1 << 22):
        d, idx, status, info = fused_scan(hip_device, ds4, q, 512, 20, r_offset=off, flags=OVERLAP)
        assert info["path"] == 3 and status[0] == 0
        oi = oidx.copy(); oi[..., 0] += off
        assert_exact(d, idx, od, oi, f"overlap ranking, row offset {off}")


def test_sharded_class_on_private_streams(hip_device, oracle_mod, tmp_path):
    """ShardedPathShadowing(streams=3): consecutive scan_begin() calls on private streams as overlap launches, the exchange
    (1-rank RCCL: all-gather + merge still run) behind each; pipelined and checked."""
    import torch.distributed as dist
    import shadowing_amd as sa
    from shadowing_amd.distributed import ShardedPathShadowing
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1, device_id=hip_device)
    objs = []
    try:
        big = syn.dataset(16384, 2048, 5600)
        for exchange in ("library", "torch"):
            obj = ShardedPathShadowing(sa.Identity(20), sa.RelativeMSE(), big, 0, sa.PredictionContext(20), device=hip_device,
                                       always_exchange=True, exchange=exchange, streams=3)
            objs.append(obj)
            qs = [torch.tensor(syn.gbm_log_returns((1, 20), 5601 + i)) for i in range(7)]
            outs, pend = [], None
            for qi in qs:
                nxt = obj.scan_begin(qi, 256, check=False)
                if pend is not None:
                    outs.append(pend.finish())
                pend = nxt
            outs.append(pend.finish())
            torch.cuda.synchronize()
            for qi, (dd, ii) in zip(qs, outs):
                od, oi = oracle_mod.scan_topk(big, qi.numpy(), 256, h=20)
                assert_exact(dd.cpu().numpy(), ii.cpu().numpy(), od, oi, f"private streams, {exchange} exchange")
            d, paths, idx = obj.shadow(syn.rolling_queries(3, 20, 5610), 128)     # a batch through the same object
            od, opaths, oidx = oracle_mod.shadow(big, syn.rolling_queries(3, 20, 5610), 128, 20)
            assert_exact(d, idx, od, oidx, "batch on private streams")
            assert np.array_equal(paths, opaths)
    finally:
        for o in objs:
            o.close()
        dist.destroy_process_group()


def test_sharded_class_falls_back_when_the_communicator_cannot_be_created(hip_device, oracle_mod, tmp_path, monkeypatch):
    """A rank whose library cannot create its RCCL communicator (RCCL not to be opened, ncclCommInitRank failing on first contact
    with a node's links): the ranks agree through the torch.distributed group, every one drops to torch's all-gather with ONE
    warning, and the results are the oracle's -- pipelined device queries (the lean ring must NOT be used) and a checked batch.
    exchange="library" raises instead."""
    import warnings
    import torch.distributed as dist
    import shadowing_amd as sa
    from shadowing_amd import _native
    from shadowing_amd.distributed import ShardedPathShadowing

    class Broken:
        def __init__(self, *a, **kw):
            raise _native.NativeLibraryError("psh_comm_create: simulated failure")
    monkeypatch.setattr(_native, "Comm", Broken)
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1, device_id=hip_device)
    objs = []
    try:
        big = syn.dataset(8192, 2048, 5650)
        obj = ShardedPathShadowing(sa.Identity(20), sa.RelativeMSE(), big, 0, sa.PredictionContext(20), device=hip_device,
                                   always_exchange=True, streams=3)
        objs.append(obj)
        qs = [torch.tensor(syn.gbm_log_returns((1, 20), 5651 + i)).to(hip_device) for i in range(5)]
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            outs = []
            for qi in qs:
                dd, ii = obj.scan_begin(qi, 200, check=False).finish()
                outs.append((dd.clone(), ii.clone()))
            torch.cuda.synchronize()
        assert sum("communicator could not be created" in str(w.message) for w in rec) == 1, [str(w.message) for w in rec]
        assert obj.exchange == "torch" and not obj._fast
        for qi, (dd, ii) in zip(qs, outs):
            od, oi = oracle_mod.scan_topk(big, qi.cpu().numpy(), 200, h=20)
            assert_exact(dd.cpu().numpy(), ii.cpu().numpy(), od, oi, "fallback exchange")
        d, paths, idx = obj.shadow(syn.rolling_queries(3, 20, 5660), 128)
        od, opaths, oidx = oracle_mod.shadow(big, syn.rolling_queries(3, 20, 5660), 128, 20)
        assert_exact(d, idx, od, oidx, "fallback exchange, batch")
        assert np.array_equal(paths, opaths)
        strict = ShardedPathShadowing(sa.Identity(20), sa.RelativeMSE(), big, 0, sa.PredictionContext(20), device=hip_device,
                                      always_exchange=True, exchange="library")
        objs.append(strict)
        with pytest.raises(_native.NativeLibraryError):
            strict.shadow(syn.rolling_queries(1, 20, 5661), 64)
    finally:
        for o in objs:
            o.close()
        dist.destroy_process_group()


def test_sharded_class_lean_steps(hip_device, oracle_mod, tmp_path):
    """The lean form of a step (ShardedPathShadowing._fast_step: prepared argument lists, a ring of buffers, two ctypes
    calls): device queries, check=False, library exchange -- 20 single queries pipelined three deep, then a batch of 4."""
    import torch.distributed as dist
    import shadowing_amd as sa
    from shadowing_amd.distributed import ShardedPathShadowing
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1, device_id=hip_device)
    obj = None
    try:
        big = syn.dataset(16384, 2048, 5700)
        obj = ShardedPathShadowing(sa.Identity(20), sa.RelativeMSE(), big, 0, sa.PredictionContext(20), device=hip_device,
                                   always_exchange=True, exchange="library", streams=3)
        for B, n in ((1, 20), (4, 5)):
            qs = [torch.tensor(syn.gbm_log_returns((B, 20), 5701 + 10 * B + i)).to(hip_device) for i in range(n)]
            outs, pend = [], []
            for qi in qs:
                pend.append(obj.scan_begin(qi, 256, check=False))
                if len(pend) == 3:
                    dd, ii = pend.pop(0).finish()
                    outs.append((dd.clone(), ii.clone()))            # ring buffers: copy before they are reused
            for p_ in pend:
                dd, ii = p_.finish()
                outs.append((dd.clone(), ii.clone()))
            torch.cuda.synchronize()
            assert (B, 20, 256) in obj._fast, "the lean path must have served these calls"
            assert int(obj.last_status.max().item()) == 0
            for qi, (dd, ii) in zip(qs, outs):
                od, oi = oracle_mod.scan_topk(big, qi.cpu().numpy(), 256, h=20)
                assert_exact(dd.cpu().numpy(), ii.cpu().numpy(), od, oi, f"lean steps B={B}")
    finally:
        if obj is not None:
            obj.close()
        dist.destroy_process_group()


def test_cu_masked_stream(hip_device, oracle_mod):
    """psh_stream_create_reserving: a stream that leaves PSH_STREAM_RESERVED_CUS compute units alone; overlap launches issued
    on it with PSH_FLAG_RESERVE_CUS (grid = the compute units the stream may use) stay exact."""
    from shadowing_amd import _native
    s, reserved = _native.reserving_stream(hip_device)
    assert reserved == _native.PSH_STREAM_RESERVED_CUS
    ds = syn.dataset(16384, 2048, 5800)
    q = syn.gbm_log_returns((1, 20), 5801)
    ws = _native.Workspace(hip_device)
    main = torch.cuda.Stream(hip_device)                     # (masked streams are blocking streams: stay off the default stream)
    with torch.cuda.stream(main):
        ds_t = torch.as_tensor(ds[:, 0, :].copy()).to(hip_device)
        q_t = torch.as_tensor(q).to(hip_device)
    main.synchronize()
    info = {}
    with torch.cuda.stream(s):
        d, idx, st = _native.scan_topk(ds_t, q_t, 400, h=20, workspace=ws, flags=OVERLAP | _native.FLAG_RESERVE_CUS, info=info)
    s.synchronize()
    assert info["path"] == 3 and int(st[0]) == 0
    ncu = torch.cuda.get_device_properties(hip_device).multi_processor_count
    assert info["grid_blocks"] == ncu - reserved
    od, oidx = oracle_mod.scan_topk(ds, q, 400, h=20)
    assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, "overlap launches on a CU-masked stream")


@pytest.mark.parametrize("B", [2, 3])
@pytest.mark.parametrize("R,T,W,h,k", [
    (8192, 4096, 20, 20, 1024),
    (6000, 2048, 20, 11, 700),
    (3000, 2051, 20, 20, 300),      # unaligned rows
    (5000, 1100, 8, 7, 200),        # run-time window lengths
    (5000, 1100, 33, 7, 200),
    (1200, 9000, 20, 20, 64),
])
def test_two_and_three_queries_ride_the_overlap_launches(hip_device, oracle_mod, B, R, T, W, h, k):
    """Two or three queries: ONE sample launch (a sampled unit serves every query), ONE scan (the unit converted once, the
    window energies one product, four MFMAs and a threshold test per query under ONE common f16 scale), a ranking launch
    with a row of blocks per query -- the default for such batches, no flag needed.  Queries of very different magnitudes
    share the scale of the most demanding one."""
    ds = syn.dataset(R, T, 6000 + R)
    q = syn.gbm_log_returns((B, W), 6100 + W + B)
    q[B - 1] *= np.float32(7.5)                           # a louder query: it dictates the common scale
    d, idx, status, info = fused_scan(hip_device, ds, q, k, h)
    assert info["path"] == 3 and (status == 0).all()
    od, oidx = oracle_mod.scan_topk(ds, q, k, h=h)
    assert_exact(d, idx, od, oidx, f"{B} queries on the overlap launches")
    # the batched scan (what larger batches take; its own status protocol may send a query to the exhaustive path) agrees
    from shadowing_amd import _native
    ds_t = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :])).to(hip_device)
    d2, idx2 = _native.scan_topk_checked(ds_t, torch.as_tensor(q).to(hip_device), k, h=h, flags=16)      # PSH_FLAG_NO_FUSE
    torch.cuda.synchronize()
    assert_exact(d, idx, d2.cpu().numpy(), idx2.cpu().numpy(), "overlap launches vs the batched scan")


def test_small_batch_status_protocol_and_reference_golden(hip_device, oracle_mod):
    """A batch of 3 where one query is all zeros (every distance +inf: its sample carries no level -> RETRY for the step) goes
    through scan_topk_checked; and the reference's own multi-query golden through the seam."""
    from shadowing_amd import _native
    import shadowing_amd as sa
    ds = syn.dataset(8192, 2048, 6200)
    q = syn.gbm_log_returns((3, 20), 6201)
    q[1] = 0.0
    ds_t = torch.as_tensor(ds[:, 0, :].copy()).to(hip_device)
    d, idx = _native.scan_topk_checked(ds_t, torch.as_tensor(q).to(hip_device), 100, h=20)
    torch.cuda.synchronize()
    od, oidx = oracle_mod.scan_topk(ds, q, 100, h=20)
    got_d, got_i = d.cpu().numpy(), idx.cpu().numpy()
    assert np.array_equal(got_d[[0, 2]].view(np.uint32), od[[0, 2]].view(np.uint32)) and np.array_equal(got_i[[0, 2]], oidx[[0, 2]])
    assert np.isinf(got_d[1]).all()
    g = load_golden("cfg3_rolling_R2048")
    obj = sa.PathShadowing(sa.Identity(g["W"]), sa.RelativeMSE(), g["dataset"], sa.PredictionContext(g["h"]), cache=True)
    dd, paths, ii = obj.shadow(g["queries"][:3], k=g["k"], cuda=True)
    assert_matches_reference(dd, ii, {**g, "d": g["d"][:3], "idx": g["idx"][:3]}, None, what="3 queries of cfg3_rolling_R2048")


@pytest.mark.parametrize("B", [2, 3])
def test_plain_small_batch_call_reports_retry_and_the_protocol_recovers(hip_device, oracle_mod, B):
    """include/psh.h, psh_scan_topk's status protocol: a PLAIN call (no flag) with 2 or 3 queries rides the overlap
    launches and may say PSH_STATUS_RETRY for the whole call.  An ensemble of constant rows forces it (every window ties: a
    block's list overflows): the raw call must SAY so in every query's status word; a batch with an all-zero query (every
    distance of that query +inf) is served or refused -- whichever, a status of OK means exact results; and the same call
    through the protocol (PSH_FLAG_NO_FUSE, the exhaustive path where that says OVERFLOW) returns the oracle's answer."""
    from shadowing_amd import _native
    k, h = 100, 20
    cases = []
    ds = syn.dataset(8192, 2048, 6400 + B)
    q = syn.gbm_log_returns((B, 20), 6410 + B)
    q[B - 1] = 0.0
    cases.append(("zero query", ds, q))
    flat = np.full((2048, 1, 512), 0.01, np.float32)
    cases.append(("constant rows", flat, syn.gbm_log_returns((B, 20), 6420 + B)))
    for name, dsx, qx in cases:
        ds_t = torch.as_tensor(np.ascontiguousarray(dsx[:, 0, :])).to(hip_device)
        q_t = torch.as_tensor(qx).to(hip_device)
        ws = _native.Workspace(hip_device)
        info = {}
        rd, ri, st = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws, info=info)
        torch.cuda.synchronize()
        assert info.get("path") == 3, f"{name}: a plain {B}-query call takes the overlap launches (path {info.get('path')})"
        od, oidx = oracle_mod.scan_topk(dsx, qx, k, h=h)
        stat = st.cpu().numpy()
        if name == "constant rows":
            assert (stat == _native.PSH_STATUS_RETRY).all(), f"{name}: status {stat}"
            # a call that says RETRY leaves no plausible numbers behind (ABI version 2): NaN distances, (-1, -1) indices
            assert np.isnan(rd.cpu().numpy()).all() and (ri.cpu().numpy() == -1).all()
            # ... and so does the fused single launch (one query of the same ensemble), through results a good call left there
            good_ds = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :512])).to(hip_device)
            o_d = torch.empty((1, k), dtype=torch.float32, device=hip_device); o_i = torch.empty((1, k, 2), dtype=torch.int32, device=hip_device)
            ws1 = _native.Workspace(hip_device)
            _, _, st1 = _native.scan_topk(good_ds, q_t[:1].contiguous(), k, h=h, workspace=ws1, out=(o_d, o_i))
            torch.cuda.synchronize()
            assert int(st1[0]) == 0 and torch.isfinite(o_d).all()
            inf1 = {}
            _, _, st1 = _native.scan_topk(ds_t, q_t[:1].contiguous(), k, h=h, workspace=ws1, out=(o_d, o_i), info=inf1)
            torch.cuda.synchronize()
            assert inf1["path"] == 2 and int(st1[0]) == _native.PSH_STATUS_RETRY
            assert torch.isnan(o_d).all() and bool((o_i == -1).all())
        else:
            assert (stat == _native.PSH_STATUS_RETRY).all() or (stat == _native.PSH_STATUS_OK).all(), f"{name}: status {stat}"
            if (stat == _native.PSH_STATUS_OK).all():
                # (the all-zero query: every distance +inf, any k windows are a correct answer -- the reference's order among
                #  exactly tied distances is arbitrary as well)
                live = list(range(B - 1))
                assert_exact(rd.cpu().numpy()[live], ri.cpu().numpy()[live], od[live], oidx[live], f"{B} queries, {name}, raw call with status OK")
                assert np.isinf(rd.cpu().numpy()[B - 1]).all()
        d, idx = _native.scan_topk_checked(ds_t, q_t, k, h=h, workspace=ws)
        torch.cuda.synchronize()
        if name == "zero query":
            got_d, got_i = d.cpu().numpy(), idx.cpu().numpy()
            live = list(range(B - 1))
            assert_exact(got_d[live], got_i[live], od[live], oidx[live], f"{B} queries, {name}, through the status protocol")
            assert np.isinf(got_d[B - 1]).all()
            continue
        assert_exact(d.cpu().numpy(), idx.cpu().numpy(), od, oidx, f"{B} queries, {name}, through the status protocol")


def test_shadow_async_equals_shadow(hip_device, oracle_mod):
    """PathShadowing.shadow_async(): a dozen independent queries enqueued back to back (three private streams, overlap
    launches), collected afterwards: each triple equals the blocking shadow(cuda=True)'s -- distances, gathered paths,
    indices -- and the oracle's; configurations the asynchronous path does not cover come back through the same handle."""
    import shadowing_amd as sa
    ds = syn.dataset(16384, 2048, 5900)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(horizon=20))
    qs = [syn.gbm_log_returns((20,), 5901 + i) for i in range(12)]
    handles = [obj.shadow_async(q, k=300) for q in qs]
    assert obj.last_path == "hip" and len({id(h) for h in handles}) == 12
    for q, hnd in zip(qs, handles):
        d, paths, idx = hnd.result()
        assert hnd.done() and hnd.result()[0] is d
        d0, p0, i0 = obj.shadow(q, k=300, cuda=True)
        assert np.array_equal(d.view(np.uint32), d0.view(np.uint32)) and np.array_equal(idx, i0) and np.array_equal(paths, p0)
        od, opaths, oidx = oracle_mod.shadow(ds, q[None, :], 300, 20)
        assert_exact(d, idx, od, oidx, "shadow_async")
        assert np.array_equal(paths, opaths)
    # a batch, and a Foveal embedding: served by shadow() at call time, same handle type
    qb = syn.rolling_queries(5, 20, 5950)
    d, paths, idx = obj.shadow_async(qb, k=64).result()
    d0, p0, i0 = obj.shadow(qb, k=64, cuda=True)
    assert np.array_equal(d, d0) and np.array_equal(idx, i0) and np.array_equal(paths, p0)
    fov = sa.PathShadowing(sa.Foveal(1.4, 0.9, 40), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(horizon=20))
    x = syn.gbm_log_returns((40,), 5960)
    d, paths, idx = fov.shadow_async(x, k=50).result()
    d0, p0, i0 = fov.shadow(x, k=50, cuda=True)
    assert np.array_equal(d, d0) and np.array_equal(idx, i0)
    with pytest.raises(Exception):
        obj.shadow_async(np.zeros(7, np.float32), k=5)


def test_blocking_shadow_prepared_slot(hip_device, oracle_mod):
    """shadow(cuda=True) for ONE Identity query runs through a prepared slot (fused launch on the caller's stream, results
    into buffers the object keeps): the returned arrays are the caller's own (a later call does not change them), the slot
    follows k / the ensemble / the workspace buffer when they change, a multi-channel ensemble gathers every channel, and
    data the single launch gives up on (constant rows: every window ties) comes back exact through the general path."""
    import shadowing_amd as sa
    ds = syn.dataset(8192, 2048, 6100)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(horizon=20))
    kept = []
    for i, k in enumerate((200, 200, 64, 200)):
        q = syn.gbm_log_returns((20,), 6101 + i)
        d, paths, idx = obj.shadow(q, k=k, cuda=True)
        assert obj.last_path == "hip" and getattr(obj, "_sync_slot", None) is not None
        od, opaths, oidx = oracle_mod.shadow(ds, q[None, :], k, 20)
        assert_exact(d, idx, od, oidx, f"prepared shadow call {i}")
        assert np.array_equal(paths, opaths)
        kept.append((d, paths, idx, od, opaths, oidx))
        if i == 1:
            # a batch through the same object grows its workspace: the slot is rebuilt against the new buffer
            qb = syn.rolling_queries(16, 20, 6150)
            db, _, ib = obj.shadow(qb, k=200, cuda=True)
            odb, oib = oracle_mod.scan_topk(ds, qb, 200, h=20)
            assert_exact(db, ib, odb, oib, "batch between prepared calls")
    for d, paths, idx, od, opaths, oidx in kept:                      # earlier results are untouched by later calls
        assert np.array_equal(d.view(np.uint32), od.view(np.uint32)) and np.array_equal(idx, oidx) and np.array_equal(paths, opaths)
    # another ensemble behind the same object
    ds2 = syn.dataset(4096, 1024, 6200)
    obj.dataset = torch.as_tensor(ds2)
    q = syn.gbm_log_returns((20,), 6201)
    d, paths, idx = obj.shadow(q, k=100, cuda=True)
    od, opaths, oidx = oracle_mod.shadow(ds2, q[None, :], 100, 20)
    assert_exact(d, idx, od, oidx, "prepared shadow, new ensemble")
    assert np.array_equal(paths, opaths)
    # every window ties: RETRY inside the launch -> general path, still exact
    flat = np.full((2048, 1, 512), 0.01, np.float32)
    objf = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), torch.as_tensor(flat), sa.PredictionContext(horizon=20))
    q = syn.gbm_log_returns((20,), 6300)
    for _ in range(2):
        d, paths, idx = objf.shadow(q, k=300, cuda=True)
        od, opaths, oidx = oracle_mod.shadow(flat, q[None, :], 300, 20)
        assert_exact(d, idx, od, oidx, "prepared shadow, ties")
        assert np.array_equal(paths, opaths)


def test_blocking_shadow_with_admission_hints(hip_device, oracle_mod):
    """PathShadowing(hint="auto"): consecutive rolling query dates hand the library the level the previous date's k-th distance
    predicts (psh_profile.tau_hint -> the fused launch without its sample phase and first barrier).  Results are the exact
    top-k whether a hint holds ("ok"), falls short ("short": one more launch without it, hints off for a few calls) or is
    fooled on purpose."""
    import shadowing_amd as sa
    ds = syn.dataset(8192, 2048, 6400)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(horizon=20), hint="auto")
    qs = syn.rolling_queries(24, 20, 6401)
    seen = []
    for i in range(24):
        d, paths, idx = obj.shadow(qs[i], k=200, cuda=True)
        assert obj.last_path == "hip"
        od, opaths, oidx = oracle_mod.shadow(ds, qs[i][None, :], 200, 20)
        assert_exact(d, idx, od, oidx, f"hinted call {i} ({obj.last_hint})")
        assert np.array_equal(paths, opaths)
        seen.append(obj.last_hint)
    assert seen[0] is None and "ok" in seen                       # the first call has nothing to go by; some hints held
    # a fooled hint (the previous k-th distance a tenth of what it was: far fewer than k windows below the level)
    obj._hint_state["dk"] *= 0.1
    obj._hint_state["skip"] = 0
    d, paths, idx = obj.shadow(qs[5], k=200, cuda=True)
    assert obj.last_hint == "short" and obj._hint_state["skip"] > 0
    od, opaths, oidx = oracle_mod.shadow(ds, qs[5][None, :], 200, 20)
    assert_exact(d, idx, od, oidx, "fooled hint (too low)")
    # ... and one far too generous (more candidates than the blocks' lists hold): same recovery
    obj._hint_state["dk"] *= 3.0
    obj._hint_state["skip"] = 0
    d, paths, idx = obj.shadow(qs[6], k=200, cuda=True)
    assert obj.last_hint == "short"
    od, opaths, oidx = oracle_mod.shadow(ds, qs[6][None, :], 200, 20)
    assert_exact(d, idx, od, oidx, "fooled hint (too high)")
    # without hint="auto" nothing is hinted
    plain = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), torch.as_tensor(ds), sa.PredictionContext(horizon=20))
    plain.shadow(qs[0], k=200, cuda=True); plain.shadow(qs[1], k=200, cuda=True)
    assert plain.last_hint is None and plain._hint_state is None


def test_cuda_dataset_is_scanned_for_nonfinite_samples_once(hip_device, oracle_mod, monkeypatch):
    """A 2-D (R, T) CUDA tensor as the dataset: _dim_array hands a fresh (R, 1, T) view on every call -- the pass over the
    ensemble that looks for NaN / inf (plus its host synchronisation) must still run once per storage and version, not once
    per shadow() call (the cache keyed on the view object never hit)."""
    import shadowing_amd as sa
    from shadowing_amd import _native
    ds = syn.dataset(4096, 1024, 6500)
    dev_ds = torch.as_tensor(ds[:, 0, :]).to(hip_device)               # (R, T) on the device
    counted = []
    real = _native.count_nonfinite
    monkeypatch.setattr(_native, "count_nonfinite", lambda t: (counted.append(1), real(t))[1])
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), dev_ds, sa.PredictionContext(horizon=20))
    for i in range(4):
        q = syn.gbm_log_returns((20,), 6501 + i)
        d, _, idx = obj.shadow(q, k=50, cuda=True)
        od, _, oidx = oracle_mod.shadow(ds, q[None, :], 50, 20)
        assert_exact(d, idx, od, oidx, f"call {i}")
    obj.shadow(syn.rolling_queries(5, 20, 6510), k=50, cuda=True)      # (the batched path and shadow_async share the cache)
    obj.shadow_async(syn.gbm_log_returns((20,), 6511), k=50).result()
    assert len(counted) == 1
    dev_ds[7, 100] = float("nan")                                       # an in-place edit bumps the version: looked at again
    q = syn.gbm_log_returns((20,), 6512)
    d, _, idx = obj.shadow(q, k=50, cuda=True)
    assert len(counted) == 2 and obj._dirty
    ds2 = ds.copy(); ds2[7, 0, 100] = np.nan
    od, _, oidx = oracle_mod.shadow(ds2, q[None, :], 50, 20)
    assert_exact(d, idx, od, oidx, "after the edit")
