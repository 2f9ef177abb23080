"""ctypes binding of libpsh_hip.so (include/psh.h) on PyTorch-ROCm tensors.

PyTorch is plumbing here: device memory, streams, the caching allocator.  All
compute goes through the C ABI.  There is NO CPU fallback in this module: if the
library cannot be loaded, or a tensor is not on a HIP device, calls raise.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch

from . import _build

PSH_OK = 0
PSH_VERSION = 3          # include/psh.h: 2: psh_profile.tau_hint, psh_candidates_layout; 3: psh_shadow_blocking
PSH_STATUS_OK, PSH_STATUS_OVERFLOW, PSH_STATUS_RETRY = 0, 1, 2
PSH_MAX_W, PSH_MAX_K, PSH_MAX_B_PER_LAUNCH = 256, 16384, 1024
# psh_profile.flags (include/psh.h)
FLAG_UNSORTED, FLAG_FILTER_VALU, FLAG_EMBED_DENSE, FLAG_ROWS_GENERIC, FLAG_NO_FUSE, FLAG_RESERVE_CUS, FLAG_EMBED_MX, FLAG_EMBED_TAPS, FLAG_EMBED_PLAN_KEEP, FLAG_EMBED_MX_SPLIT, FLAG_SELECT_ONE_BLOCK, FLAG_OVERLAP, FLAG_MQ_F16, FLAG_LONG_LOOP = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192


class NativeLibraryError(RuntimeError):
    """The HIP extension is missing or failed."""


class PshProfile(C.Structure):
    _fields_ = [("mode", C.c_int), ("flags", C.c_int), ("ev_scan_begin", C.c_void_p), ("ev_scan_end", C.c_void_p),
                ("prep_ms", C.c_float), ("sample_ms", C.c_float), ("threshold_ms", C.c_float),
                ("scan_ms", C.c_float), ("select_ms", C.c_float), ("total_ms", C.c_float),
                ("path", C.c_int), ("n_sample_rows", C.c_int), ("grid_blocks", C.c_int),
                ("n_candidates", C.c_int), ("tau_hint", C.c_void_p)]

    def as_dict(self) -> dict:
        skip = ("mode", "flags", "ev_scan_begin", "ev_scan_end", "tau_hint")
        return {name: getattr(self, name) for name, _ in self._fields_ if name not in skip}


EXPORTS = ("psh_version", "psh_strerror", "psh_last_hip_error", "psh_workspace_bytes", "psh_query_norm",
           "psh_scan_topk", "psh_scan_topk_exhaustive", "psh_scan_topk_embedded",
           "psh_scan_topk_embedded_exhaustive", "psh_merge_workspace_bytes", "psh_merge_topk",
           "psh_merge_topk_gathered", "psh_merge_sorted_gathered", "psh_gather_paths", "psh_embed_rows",
           "psh_embedded_supported", "psh_embed_plan_offset", "psh_candidates_layout", "psh_workspace_init", "psh_last_comm_error", "psh_comm_unique_id", "psh_comm_create",
           "psh_comm_destroy", "psh_comm_world", "psh_exchange_merge", "psh_stream_create_reserving", "psh_stream_destroy",
           "psh_weighted_moments", "psh_realized_variance", "psh_count_nonfinite", "psh_smear_nonfinite", "psh_rows_nonfinite",
           "psh_shadow_block_layout", "psh_shadow_blocking")

_lib = None


def library_path() -> Path:
    import os
    override = os.environ.get("PSH_LIB")          # tuning aid: load an experimental build
    return Path(override) if override else _build.LIB


def load() -> C.CDLL:
    """dlopen the in-tree library (never builds implicitly on a GPU box: the built
    .so travels with the tree; __graft_entry__.build() / `python -m shadowing_amd._build`
    produce it)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise NativeLibraryError(
            f"{path} is missing: build it with `python -m shadowing_amd._build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for cuda=True.")
    if path == _build.LIB and _build.is_stale():
        raise NativeLibraryError(
            f"{path} was built from other sources than the ones in shadowing_amd/csrc (content hash differs): "
            "rebuild it with `python -m shadowing_amd._build`")
    try:
        L = C.CDLL(str(path))
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {path}: {e}") from e
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.psh_version.restype = i32
    L.psh_strerror.restype = C.c_char_p
    L.psh_strerror.argtypes = [i32]
    L.psh_last_hip_error.restype = C.c_char_p
    L.psh_workspace_bytes.restype = i32
    L.psh_workspace_bytes.argtypes = [i64, i64, i32, i32, i32, i32, C.POINTER(C.c_size_t)]
    L.psh_last_comm_error.restype = C.c_char_p
    L.psh_comm_unique_id.restype = i32
    L.psh_comm_unique_id.argtypes = [C.c_char_p, vp]
    L.psh_comm_create.restype = i32
    L.psh_comm_create.argtypes = [C.c_char_p, i32, i32, i32, vp, C.POINTER(vp)]
    L.psh_comm_destroy.restype = i32
    L.psh_comm_destroy.argtypes = [vp]
    L.psh_comm_world.restype = i32
    L.psh_comm_world.argtypes = [vp]
    L.psh_exchange_merge.restype = i32
    L.psh_exchange_merge.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, C.c_size_t, vp, vp]
    L.psh_stream_create_reserving.restype = i32
    L.psh_stream_create_reserving.argtypes = [i32, i32, C.POINTER(vp), C.POINTER(i32)]
    L.psh_stream_destroy.restype = i32
    L.psh_stream_destroy.argtypes = [i32, vp]
    L.psh_workspace_init.restype = i32
    L.psh_workspace_init.argtypes = [i32, vp, vp, C.c_size_t]
    L.psh_query_norm.restype = i32
    L.psh_query_norm.argtypes = [i32, vp, vp, i32, i32, vp]
    scan_args = [i32, vp, vp, i64, i64, i64, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, C.c_size_t,
                 C.POINTER(PshProfile)]
    L.psh_scan_topk.restype = i32
    L.psh_scan_topk.argtypes = scan_args
    L.psh_scan_topk_exhaustive.restype = i32
    L.psh_scan_topk_exhaustive.argtypes = scan_args
    emb_args = [i32, vp, vp, i64, i64, i64, vp, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, C.c_size_t,
                C.POINTER(PshProfile)]
    L.psh_scan_topk_embedded.restype = i32
    L.psh_scan_topk_embedded.argtypes = emb_args
    L.psh_scan_topk_embedded_exhaustive.restype = i32
    L.psh_scan_topk_embedded_exhaustive.argtypes = emb_args
    L.psh_merge_workspace_bytes.restype = i32
    L.psh_merge_workspace_bytes.argtypes = [i32, i32, C.POINTER(C.c_size_t)]
    L.psh_merge_topk.restype = i32
    L.psh_merge_topk.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, C.c_size_t]
    L.psh_merge_topk_gathered.restype = i32
    L.psh_merge_topk_gathered.argtypes = [i32, vp, vp, vp, i32, i64, i64, i32, i32, i32, vp, vp, vp, C.c_size_t]
    L.psh_merge_sorted_gathered.restype = i32
    L.psh_merge_sorted_gathered.argtypes = [i32, vp, vp, vp, i32, i64, i64, i32, i32, i32, vp, vp]
    L.psh_embed_plan_offset.restype = C.c_size_t
    L.psh_embed_plan_offset.argtypes = []
    L.psh_candidates_layout.restype = i32
    L.psh_candidates_layout.argtypes = [i64, i64, i32, i32, i32, i32, C.c_size_t, C.POINTER(C.c_int64)]
    L.psh_embedded_supported.restype = i32
    L.psh_embedded_supported.argtypes = [i32, i32]
    L.psh_embed_rows.restype = i32
    L.psh_embed_rows.argtypes = [i32, vp, vp, i64, i64, vp, i32, i32, vp]
    L.psh_count_nonfinite.restype = i32
    L.psh_count_nonfinite.argtypes = [i32, vp, vp, i64, vp]
    L.psh_shadow_block_layout.restype = i32
    L.psh_shadow_block_layout.argtypes = [i32, i32, i32, i64, C.POINTER(C.c_size_t)]
    L.psh_shadow_blocking.restype = i32
    L.psh_shadow_blocking.argtypes = [i32, vp, vp, i64, i64, i64, vp, i64, i32, i32, i32, vp, C.c_size_t, i32, vp, C.c_size_t,
                                      C.POINTER(PshProfile)]
    L.psh_rows_nonfinite.restype = i32
    L.psh_rows_nonfinite.argtypes = [i32, vp, vp, i64, i64, i64, vp]
    L.psh_smear_nonfinite.restype = i32
    L.psh_smear_nonfinite.argtypes = [i32, vp, vp, i64, i64, i64, i32, i32, vp]
    L.psh_weighted_moments.restype = i32
    L.psh_weighted_moments.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp, vp]
    L.psh_realized_variance.restype = i32
    L.psh_realized_variance.argtypes = [i32, vp, vp, i64, i64, i32, C.POINTER(C.c_int), i32, i32, vp]
    L.psh_gather_paths.restype = i32
    L.psh_gather_paths.argtypes = [i32, vp, vp, i64, i64, i64, i64, vp, i64, i32, vp]
    if L.psh_version() != PSH_VERSION:
        raise NativeLibraryError(f"{path} speaks version {L.psh_version()} of the C ABI, this binding version {PSH_VERSION} "
                                 "(psh_profile differs): rebuild it with `python -m shadowing_amd._build`")
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc == PSH_OK:
        return
    L = load()
    msg = L.psh_strerror(rc).decode()
    if rc == -4:
        msg += ": " + L.psh_last_hip_error().decode()
    if rc == -5:
        msg += ": " + L.psh_last_comm_error().decode()
    if rc == -1:
        raise ValueError(f"{what}: {msg}")
    raise NativeLibraryError(f"{what}: {msg} (code {rc})")


def _dev_tensor(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NativeLibraryError(f"{name} must be a tensor on a HIP device (got {type(t).__name__} "
                                 f"on {getattr(t, 'device', '?')}); there is no CPU path here")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def workspace_bytes(R: int, T: int, B: int, W: int, h: int, k: int) -> int:
    out = C.c_size_t(0)
    _check(load().psh_workspace_bytes(R, T, B, W, h, k, C.byref(out)), "psh_workspace_bytes")
    return int(out.value)


class Workspace:
    """Caller-owned scratch of the scan (a torch uint8 tensor), grown on demand.  A fresh buffer is armed for the
    fused single-launch scan (psh_workspace_init, enqueued on the current stream).  The header at its start keeps an
    epoch from launch to launch: one Workspace serves one stream at a time."""

    def __init__(self, device: torch.device):
        self.device = device
        self.buf = None
        self.plan_of = None        # (kernel tensor, its version, B): whose plan the last embedded call left in `buf`

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.plan_of = None
            self.arm()
        return self.buf

    def arm(self) -> None:
        """(Re-)initialise the fused scan's header: after allocation, and after a PSH_STATUS_RETRY (a time-out inside
        the fused launch disarms the header on the device)."""
        if self.buf is not None:
            _check(load().psh_workspace_init(self.device.index, _stream_ptr(self.device), self.buf.data_ptr(),
                                             self.buf.numel()), "psh_workspace_init")


def query_norm(queries: torch.Tensor) -> torch.Tensor:
    q = _dev_tensor(queries, torch.float32, "queries")
    B, W = q.shape
    out = torch.empty(B, dtype=torch.float32, device=q.device)
    _check(load().psh_query_norm(q.device.index, _stream_ptr(q.device), q.data_ptr(), B, W, out.data_ptr()),
           "psh_query_norm")
    return out


def scan_topk(dataset: torch.Tensor, queries: torch.Tensor, k: int, h: int = 0, r_offset: int = 0,
              qnorm: torch.Tensor | None = None, workspace: Workspace | None = None,
              exhaustive: bool = False, profile: bool = False, extra_workspace_factor: float = 1.0,
              scan_events: tuple | None = None, out: tuple | None = None, unsorted: bool = False, flags: int = 0,
              info: dict | None = None, tau_hint: torch.Tensor | None = None):
    """Enqueue the scan on the current stream.

    dataset (R, T) float32 device, queries (B, W) float32 device.  Returns
    (d (B,k) f32, idx (B,k,2) i32, status (B,) i32[, profile dict]) -- device tensors,
    NOT synchronised; status must be inspected (after a sync) unless exhaustive=True.
    `profile=True` synchronises and adds the per-stage HIP-event timings;
    `scan_events=(begin, end)` (two torch.cuda.Event(enable_timing=True), each recorded
    once beforehand so that the handle exists) are re-recorded on the current stream
    right around the dominant scan kernel, without any synchronisation.
    `unsorted=True`: the k best come back in arbitrary order (PSH_FLAG_UNSORTED; for callers that
    merge afterwards).  `flags`: further PSH_FLAG_* bits (A/B switches of tests and tools).
    `info`: a dict that receives the launch plan's facts (path: 0 separate launches, 1 exhaustive, 2 fused, 3 the
    overlap-friendly launches; ...) without any synchronisation.
    `tau_hint`: (B,) float32 device tensor, the caller's admission levels on acc = (d ||x||)^2 (psh_profile.tau_hint): no
    bootstrap sample; a status other than OK then means "rerun without the hint".
    """
    ds = _dev_tensor(dataset, torch.float32, "dataset")
    q = _dev_tensor(queries, torch.float32, "queries")
    if ds.dim() != 2 or q.dim() != 2:
        raise ValueError("dataset must be (R, T) and queries (B, W)")
    if q.device != ds.device:
        raise ValueError("dataset and queries must live on the same device")
    R, T = ds.shape
    B, W = q.shape
    dev = ds.device
    if qnorm is not None:
        qnorm = _dev_tensor(qnorm, torch.float32, "qnorm")
    if B > PSH_MAX_B_PER_LAUNCH and not profile and scan_events is None:
        # one launch keeps a per-block append cursor per query in LDS: batch the queries
        parts = [scan_topk(ds, q[i:i + PSH_MAX_B_PER_LAUNCH].contiguous(), k, h=h, r_offset=r_offset,
                           qnorm=None if qnorm is None else qnorm[i:i + PSH_MAX_B_PER_LAUNCH].contiguous(),
                           workspace=workspace, exhaustive=exhaustive, extra_workspace_factor=extra_workspace_factor,
                           unsorted=unsorted, flags=flags,
                           tau_hint=None if tau_hint is None else tau_hint[i:i + PSH_MAX_B_PER_LAUNCH].contiguous())
                 for i in range(0, B, PSH_MAX_B_PER_LAUNCH)]
        return tuple(torch.cat([p[j] for p in parts], dim=0) for j in range(3))
    nbytes = int(workspace_bytes(R, T, B, W, h, k) * extra_workspace_factor)
    ws = (workspace or Workspace(dev)).get(nbytes)
    status = None
    if out is not None:       # caller-provided (B,k) f32 / (B,k,2) i32 contiguous device tensors (e.g. views of a send buffer)
        out_d = _dev_tensor(out[0], torch.float32, "out[0]")
        out_idx = _dev_tensor(out[1], torch.int32, "out[1]")
        if tuple(out_d.shape) != (B, k) or tuple(out_idx.shape) != (B, k, 2):
            raise ValueError("out must be ((B,k) float32, (B,k,2) int32)")
        if len(out) > 2:      # ... and optionally the (B,) int32 status words (no allocation at all in the call then)
            status = _dev_tensor(out[2], torch.int32, "out[2]")
            if tuple(status.shape) != (B,):
                raise ValueError("out[2] must be (B,) int32")
    else:
        out_d = torch.empty((B, k), dtype=torch.float32, device=dev)
        out_idx = torch.empty((B, k, 2), dtype=torch.int32, device=dev)
    if status is None:
        status = torch.empty((B,), dtype=torch.int32, device=dev)
    prof = None
    if profile:
        prof = PshProfile()
        prof.mode = 0
    elif scan_events is not None:
        prof = PshProfile()
        prof.mode = 1
        prof.ev_scan_begin = scan_events[0].cuda_event
        prof.ev_scan_end = scan_events[1].cuda_event
    if unsorted and not exhaustive:
        flags |= FLAG_UNSORTED
    if tau_hint is not None:
        tau_hint = _dev_tensor(tau_hint, torch.float32, "tau_hint")
        if tuple(tau_hint.shape) != (B,):
            raise ValueError("tau_hint must be (B,) float32")
    if flags or info is not None or tau_hint is not None:
        if prof is None:
            prof = PshProfile()
            prof.mode = 1                 # no events given: nothing is recorded, nothing is synchronised
        prof.flags = flags
        prof.tau_hint = None if tau_hint is None else tau_hint.data_ptr()
    fn = load().psh_scan_topk_exhaustive if exhaustive else load().psh_scan_topk
    rc = fn(dev.index, _stream_ptr(dev), ds.data_ptr(), R, T, r_offset, q.data_ptr(),
            None if qnorm is None else qnorm.data_ptr(), B, W, h, k,
            out_d.data_ptr(), out_idx.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(),
            C.byref(prof) if prof is not None else None)
    _check(rc, "psh_scan_topk_exhaustive" if exhaustive else "psh_scan_topk")
    if info is not None:
        info.update(path=prof.path, n_sample_rows=prof.n_sample_rows, grid_blocks=prof.grid_blocks)
    if profile:
        return out_d, out_idx, status, prof.as_dict()
    return out_d, out_idx, status


def scan_topk_checked(dataset: torch.Tensor, queries: torch.Tensor, k: int, h: int = 0, r_offset: int = 0,
                      workspace: Workspace | None = None, out: tuple | None = None, unsorted: bool = False, flags: int = 0,
                      tau_hint: torch.Tensor | None = None):
    """scan_topk + the status protocol of include/psh.h, with ONE host synchronisation in the normal case:
    a call with `tau_hint` whose status is not OK everywhere (the hint fell short of k windows, or was useless) -> the same
    call without the hint, then as below;
    PSH_STATUS_RETRY (the fused launch gave up) -> the same call through the separate launches (PSH_FLAG_NO_FUSE);
    PSH_STATUS_OVERFLOW (candidate slices overflowed: ties en masse) -> those queries through the exhaustive path.
    Returns (d, idx) device tensors holding valid results for every query."""
    ws = workspace or Workspace(dataset.device)
    d, idx, status = scan_topk(dataset, queries, k, h=h, r_offset=r_offset, workspace=ws, out=out, unsorted=unsorted, flags=flags,
                               tau_hint=tau_hint)
    st = status.cpu()
    if tau_hint is not None and bool((st != PSH_STATUS_OK).any()):
        if bool((st == PSH_STATUS_RETRY).any()):
            ws.arm()
        d, idx, status = scan_topk(dataset, queries, k, h=h, r_offset=r_offset, workspace=ws, out=out, unsorted=unsorted, flags=flags)
        st = status.cpu()
    if bool((st == PSH_STATUS_RETRY).any()):
        ws.arm()
        d, idx, status = scan_topk(dataset, queries, k, h=h, r_offset=r_offset, workspace=ws, out=out, unsorted=unsorted,
                                   flags=(flags & ~FLAG_OVERLAP) | FLAG_NO_FUSE)
        st = status.cpu()
    if bool((st != PSH_STATUS_OK).any()):      # (nothing is uploaded in the normal case)
        bad = torch.nonzero(st != PSH_STATUS_OK).flatten().to(dataset.device)
        d2, i2, _ = scan_topk(dataset, queries[bad].contiguous(), k, h=h, r_offset=r_offset, workspace=ws, exhaustive=True)
        d[bad] = d2
        idx[bad] = i2
    return d, idx


PSH_EMB_MAX_D = 128


def embed_plan(workspace: "Workspace") -> dict:
    """What the last sampled scan_topk_embedded call on `workspace` found in its kernel matrix (synchronises)."""
    off = int(load().psh_embed_plan_offset())
    v = workspace.buf[off:off + 16].view(torch.int32).cpu().tolist()
    return {"one_interval": bool(v[0]), "ktop": v[1], "merged_rows": v[2], "d": v[3]}


def embedding_supported(d: int, K: int) -> bool:
    """Whether psh_scan_topk_embedded takes a (d, K) kernel (it lives in LDS beside the wave tiles)."""
    if not (0 < d <= PSH_EMB_MAX_D and 0 < K <= PSH_MAX_W):
        return False
    return bool(load().psh_embedded_supported(int(d), int(K)))


def scan_topk_embedded(dataset: torch.Tensor, kernel: torch.Tensor, hx: torch.Tensor, k: int, h: int = 0,
                       r_offset: int = 0, hxnorm: torch.Tensor | None = None, workspace: Workspace | None = None,
                       exhaustive: bool = False, profile: bool = False, out: tuple | None = None, flags: int = 0,
                       keep_plan: bool = False, tau_hint: torch.Tensor | None = None, info: dict | None = None):
    """The scan behind a linear embedding: kernel (d, K) float32 device (unpadded), hx (B, d)
    embedded queries.  Same returns and conventions as scan_topk.
    keep_plan: the caller changes `kernel` only through torch (in-place edits bump its version): when the last sampled call
    on `workspace` scanned with this very tensor at this version, what the library found in the matrix is used again
    (PSH_FLAG_EMBED_PLAN_KEEP) instead of being worked out by one more launch."""
    ds = _dev_tensor(dataset, torch.float32, "dataset")
    ker = _dev_tensor(kernel, torch.float32, "kernel")
    q = _dev_tensor(hx, torch.float32, "hx")
    if ds.dim() != 2 or ker.dim() != 2 or q.dim() != 2 or q.shape[1] != ker.shape[0]:
        raise ValueError("dataset must be (R, T), kernel (d, K) and hx (B, d)")
    if q.device != ds.device or ker.device != ds.device:
        raise ValueError("dataset, kernel and hx must live on the same device")
    R, T = ds.shape
    d, K = ker.shape
    B = q.shape[0]
    dev = ds.device
    if hxnorm is not None:
        hxnorm = _dev_tensor(hxnorm, torch.float32, "hxnorm")
    if B > PSH_MAX_B_PER_LAUNCH and not profile:
        parts = [scan_topk_embedded(ds, ker, q[i:i + PSH_MAX_B_PER_LAUNCH].contiguous(), k, h=h, r_offset=r_offset,
                                    hxnorm=None if hxnorm is None else hxnorm[i:i + PSH_MAX_B_PER_LAUNCH].contiguous(),
                                    workspace=workspace, exhaustive=exhaustive, flags=flags, keep_plan=keep_plan,
                                    tau_hint=None if tau_hint is None else tau_hint[i:i + PSH_MAX_B_PER_LAUNCH].contiguous())
                 for i in range(0, B, PSH_MAX_B_PER_LAUNCH)]
        return tuple(torch.cat([p[j] for p in parts], dim=0) for j in range(3))
    wsobj = workspace or Workspace(dev)
    ws = wsobj.get(workspace_bytes(R, T, B, K, h, k))
    if not exhaustive:
        po = wsobj.plan_of
        planned = not (flags & (FLAG_EMBED_DENSE | FLAG_EMBED_TAPS | FLAG_EMBED_MX))     # the library works the plan out at all
        if keep_plan and planned and po is not None and po[0] is kernel and po[1:] == (kernel._version, B):
            flags |= FLAG_EMBED_PLAN_KEEP
        # (a call of the non-exhaustive entry point leaves the plan of ITS kernel in the workspace, unless it kept one)
        wsobj.plan_of = (kernel, kernel._version, B) if planned else None
    if out is not None:
        out_d = _dev_tensor(out[0], torch.float32, "out[0]")
        out_idx = _dev_tensor(out[1], torch.int32, "out[1]")
        if tuple(out_d.shape) != (B, k) or tuple(out_idx.shape) != (B, k, 2):
            raise ValueError("out must be ((B,k) float32, (B,k,2) int32)")
    else:
        out_d = torch.empty((B, k), dtype=torch.float32, device=dev)
        out_idx = torch.empty((B, k, 2), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    prof = None
    if profile:
        prof = PshProfile()
        prof.mode = 0
    if tau_hint is not None:
        tau_hint = _dev_tensor(tau_hint, torch.float32, "tau_hint")
        if tuple(tau_hint.shape) != (B,):
            raise ValueError("tau_hint must be (B,) float32")
    if flags or tau_hint is not None or info is not None:
        if prof is None:
            prof = PshProfile()
            prof.mode = 1
        prof.flags = flags
        prof.tau_hint = None if tau_hint is None else tau_hint.data_ptr()
    name = "psh_scan_topk_embedded_exhaustive" if exhaustive else "psh_scan_topk_embedded"
    rc = getattr(load(), name)(dev.index, _stream_ptr(dev), ds.data_ptr(), R, T, r_offset, ker.data_ptr(), d, K,
                               q.data_ptr(), None if hxnorm is None else hxnorm.data_ptr(), B, h, k,
                               out_d.data_ptr(), out_idx.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(),
                               C.byref(prof) if prof is not None else None)
    _check(rc, name)
    if info is not None:
        info.update(path=prof.path, n_sample_rows=prof.n_sample_rows, grid_blocks=prof.grid_blocks)
    if profile:
        return out_d, out_idx, status, prof.as_dict()
    return out_d, out_idx, status


def candidates_layout(R: int, T: int, B: int, W: int, h: int, k: int, nbytes: int) -> dict:
    """psh_candidates_layout: where a scan with these sizes on a workspace of `nbytes` leaves the windows it admitted
    (diagnostics: tests/test_gpu_admitted_set.py reads the admitted SET back and compares it with the oracle's)."""
    out = (C.c_int64 * 14)()
    _check(load().psh_candidates_layout(R, T, B, W, h, k, nbytes, out), "psh_candidates_layout")
    names = ("qstate", "bcount", "bcount2", "cand_d", "cand_rt", "cap", "hdr_cand", "hdr_blk", "hdr_stream_ncand", "max_blocks",
             "fused_max_blocks", "fused_front", "stream_list", "stream_cap")
    return dict(zip(names, (int(v) for v in out)))


def embed_rows(dataset: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """(R, d): the embedding of the FIRST window of every row (one-window rows behind a linear embedding)."""
    ds = _dev_tensor(dataset, torch.float32, "dataset")
    ker = _dev_tensor(kernel, torch.float32, "kernel")
    R, T = ds.shape
    d, K = ker.shape
    out = torch.empty((R, d), dtype=torch.float32, device=ds.device)
    _check(load().psh_embed_rows(ds.device.index, _stream_ptr(ds.device), ds.data_ptr(), R, T, ker.data_ptr(), d, K,
                                 out.data_ptr()), "psh_embed_rows")
    return out


# ---- multi-GPU exchange under the C ABI (psh_comm.hip) ----------------------------------------------------------
PSH_COMM_ID_BYTES = 128


def rccl_library_path() -> str:
    """The RCCL this process already uses: the one bundled with PyTorch-ROCm (torch/lib/librccl.so)."""
    import os
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else "librccl.so"


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(PSH_COMM_ID_BYTES)
    _check(load().psh_comm_unique_id(rccl_library_path().encode(), buf), "psh_comm_unique_id")
    return buf.raw


class Comm:
    """An RCCL communicator owned by libpsh_hip.so (psh_comm_create): one per rank, created collectively."""

    def __init__(self, device: torch.device, world: int, rank: int, unique_id: bytes):
        self.device, self.world, self.rank = device, world, rank
        h = C.c_void_p()
        _check(load().psh_comm_create(rccl_library_path().encode(), device.index, world, rank, unique_id, C.byref(h)),
               "psh_comm_create")
        self._h = h

    def close(self):
        if self._h:
            load().psh_comm_destroy(self._h)
            self._h = None

    def exchange_merge(self, send: torch.Tensor, gathered: torch.Tensor, B: int, k: int, out_d: torch.Tensor,
                       out_idx: torch.Tensor, merge_ws: torch.Tensor | None, side: "torch.cuda.Stream",
                       ev_scan_done: "torch.cuda.Event", ev_merged: "torch.cuda.Event"):
        """psh_exchange_merge: all-gather + merge on `side`, behind ONE event on the current stream; nothing is
        synchronised.  The events must have been recorded once (their handles exist)."""
        _dev_tensor(send, torch.int32, "send")
        _dev_tensor(gathered, torch.int32, "gathered")
        rc = load().psh_exchange_merge(self._h, _stream_ptr(self.device), side.cuda_stream, send.data_ptr(), gathered.data_ptr(),
                                       B, k, out_d.data_ptr(), out_idx.data_ptr(),
                                       None if merge_ws is None else merge_ws.data_ptr(), 0 if merge_ws is None else merge_ws.numel(),
                                       ev_scan_done.cuda_event, ev_merged.cuda_event)
        _check(rc, "psh_exchange_merge")


PSH_STREAM_RESERVED_CUS = 8


def reserving_stream(device: torch.device, reserve_cus: int = PSH_STREAM_RESERVED_CUS):
    """(torch stream, CUs reserved): a stream whose kernels leave `reserve_cus` compute units alone
    (psh_stream_create_reserving), wrapped for torch.  The raw stream lives as long as the process (a handful per object)."""
    h, got = C.c_void_p(), C.c_int(0)
    _check(load().psh_stream_create_reserving(device.index, int(reserve_cus), C.byref(h), C.byref(got)), "psh_stream_create_reserving")
    return torch.cuda.ExternalStream(h.value, device=device), int(got.value)


class PreparedShadow:
    """One single-query shadow() on the device -- psh_scan_topk as the overlap-friendly launches, then psh_gather_paths -- with
    everything but the stream fixed beforehand: argument lists built once, the query staged through a pinned buffer of its own,
    results in buffers the slot owns.  The per-call host cost is one small copy and two ctypes calls (the general path spends
    ~250 us of Python per call, three times what the device needs).  One slot serves one call at a time: the caller hands it
    out again only after the previous call's results have been taken."""

    def __init__(self, rows: torch.Tensor, ds3: torch.Tensor, W: int, k: int, h: int, workspace: "Workspace", flags: int,
                 host_direct: bool = False):
        """`host_direct`: the kernels read the query from the pinned staging buffer and write their results into the pinned
        result buffer themselves (host memory mapped into the device's address space: ~170 KB over PCIe as posted writes) --
        no copy engine in the chain of a BLOCKING call, whose latency is what counts."""
        self.host_direct = host_direct
        rows = _dev_tensor(rows, torch.float32, "rows")
        dev = rows.device
        R, T = rows.shape
        C_ = ds3.shape[1]
        # the query and, behind it at a 16-byte boundary, the caller's admission hint (psh_profile.tau_hint: one float)
        self._Wp = (W + 3) // 4 * 4
        self._stage_pin = torch.zeros((self._Wp + 4,), dtype=torch.float32, pin_memory=True)
        self._stage_dev = torch.zeros((self._Wp + 4,), dtype=torch.float32, device=dev)
        self.q_pin = self._stage_pin[:W].view(1, W)
        self.q_dev = self._stage_dev[:W].view(1, W)
        # results packed in ONE device buffer and one pinned host buffer (status | d | idx | paths): one D2H copy per call
        n_d, n_i, n_p = 4 * k, 8 * k, 4 * k * C_ * (W + h)
        o_d = 256
        o_i = o_d + (n_d + 255) // 256 * 256
        o_p = o_i + (n_i + 255) // 256 * 256
        self._res = torch.zeros(o_p + n_p, dtype=torch.uint8, device=dev)
        self._res_host = torch.zeros(self._res.numel(), dtype=torch.uint8, pin_memory=True)

        def carve(buf):
            return (buf[o_d:o_d + n_d].view(torch.float32).view(1, k),
                    buf[o_p:o_p + n_p].view(torch.float32).view(1, k, C_, W + h),
                    buf[o_i:o_i + n_i].view(torch.int32).view(1, k, 2),
                    buf[0:4].view(torch.int32))
        self._carve = carve
        self.host = carve(self._res_host)
        self.d, self.paths, self.idx, self.status = self.host if host_direct else carve(self._res)
        if host_direct:
            self.q_dev = self.q_pin
            self._stage_dev = self._stage_pin
        self._hint_ptr = self._stage_dev.data_ptr() + 4 * self._Wp
        self._stage_np = self._stage_pin.numpy()             # (numpy views: element stores / reads without a torch dispatch)
        self._q_np = self._stage_np[:W]
        self._refresh_status_view()
        self.event = torch.cuda.Event()
        ws = workspace.get(workspace_bytes(R, T, 1, W, h, k))
        self._keep = (rows, ds3, ws, workspace)
        self.prof = PshProfile()
        self.prof.mode = 1
        self.prof.flags = flags
        L = load()
        self._scan_fn, self._gather_fn = L.psh_scan_topk, L.psh_gather_paths
        self._scan_args = [dev.index, None, rows.data_ptr(), R, T, 0, self.q_dev.data_ptr(), None, 1, W, h, k, self.d.data_ptr(),
                           self.idx.data_ptr(), self.status.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(self.prof)]
        self._gather_args = [dev.index, None, ds3.data_ptr(), ds3.shape[0], C_, ds3.shape[2], 0, self.idx.data_ptr(), k, W + h,
                             self.paths.data_ptr()]

    def _refresh_status_view(self):
        self.status_np = self.host[3].numpy()                # the call's status word as the host sees it (after the event)

    # results of this size and more are HANDED to the caller instead of copied out of the slot's pinned buffer (host_direct
    # slots): at k = 8192 the four numpy copies were 100 us of a 340 us shadow() call
    HANDOVER_BYTES = 256 * 1024

    def take(self):
        """(d, paths, idx, status) of the finished call as numpy arrays the caller owns.  Small results are copied out of the
        pinned buffer (12 us at k = 1024); large ones keep the buffer -- the arrays view it, it lives as long as they do -- and
        the slot gets a fresh pinned buffer for its next call (torch's pinned-memory cache hands back the block a dropped
        result returned: no allocation in a steady loop)."""
        if self._res_host.numel() < self.HANDOVER_BYTES:
            return tuple(t.numpy().copy() for t in self.host)
        out = tuple(t.numpy() for t in self.host)
        self._res_host = torch.empty(self._res_host.numel(), dtype=torch.uint8, pin_memory=True)
        self.host = self._carve(self._res_host)
        self._refresh_status_view()
        if self.host_direct:                                 # the kernels write the pinned buffer themselves: new addresses
            self.host[3].zero_()
            self.d, self.paths, self.idx, self.status = self.host
            self._scan_args[12], self._scan_args[13], self._scan_args[14] = self.d.data_ptr(), self.idx.data_ptr(), self.status.data_ptr()
            self._gather_args[7], self._gather_args[10] = self.idx.data_ptr(), self.paths.data_ptr()
        return out

    def launch(self, stream: "torch.cuda.Stream", x_row: torch.Tensor, hint: float | None = None) -> None:
        """x_row: (1, W) float32 CPU tensor.  Everything is enqueued on `stream`, the D2H copies of the results included.
        `hint`: the caller's admission level on acc (psh_profile.tau_hint) -- a status other than OK then means "again without"."""
        # (a CUDA query, or one that requires grad -- the reference's `_torch` passes tensors through as they are: a detached host copy)
        self._q_np[:] = x_row.detach().cpu().numpy().reshape(-1) if isinstance(x_row, torch.Tensor) else x_row
        if hint is not None:
            self._stage_np[self._Wp] = hint
        self.prof.tau_hint = self._hint_ptr if hint is not None else None
        sp = stream.cuda_stream
        if self.host_direct:
            # nothing here is a torch operation: the kernels read the pinned staging buffer and write the pinned result buffer
            # themselves -- two library calls and the event, no stream context to enter and leave (5 us of a blocking call)
            a = self._scan_args
            a[1] = sp
            rc = self._scan_fn(*a)
            if rc:
                _check(rc, "psh_scan_topk")
            g = self._gather_args
            g[1] = sp
            rc = self._gather_fn(*g)
            if rc:
                _check(rc, "psh_gather_paths")
            self.event.record(stream)
            return
        with torch.cuda.stream(stream):
            self._stage_dev.copy_(self._stage_pin, non_blocking=True)
            a = self._scan_args
            a[1] = sp
            rc = self._scan_fn(*a)
            if rc:
                _check(rc, "psh_scan_topk")
            g = self._gather_args
            g[1] = sp
            rc = self._gather_fn(*g)
            if rc:
                _check(rc, "psh_gather_paths")
            self._res_host.copy_(self._res, non_blocking=True)
            self.event.record()


class _HostBlock:
    """One pinned block of psh_shadow_blocking (query in, status / d / idx / paths out).  `root` is the numpy view every array
    handed to a caller is based on: while the caller keeps one of them, root's reference count says so and the block is not
    written again."""

    def __init__(self, owner: "BlockingShadow"):
        import sys
        import numpy as np
        self.t = torch.zeros(owner.nbytes, dtype=torch.uint8, pin_memory=True)
        self.root = self.t.numpy()
        W = owner.W
        self.q = self.root[owner.o_query:owner.o_query + 4 * W].view(np.float32)
        self.hint = self.root[owner.o_hint:owner.o_hint + 4].view(np.float32)
        self.status = self.root[owner.o_status:owner.o_status + 4].view(np.int32)
        self.times = self.root[16:32].view(np.float32)        # PSH_SHADOW_OFF_TIMES: us enqueueing / waiting / until the launch started, last call
        self.prof = PshProfile()
        self.prof.flags = owner.flags
        rows, ds3 = owner.rows, owner.ds3
        self.args = [rows.device.index, None, rows.data_ptr(), rows.shape[0], rows.shape[1], 0, ds3.data_ptr(), ds3.shape[1],
                     W, owner.h, owner.k, self.t.data_ptr(), owner.nbytes, 0, owner.ws.data_ptr(), owner.ws.numel(), C.byref(self.prof)]
        self._getrc = sys.getrefcount
        self.rc0 = self._getrc(self.root)

    def busy(self) -> bool:
        return self._getrc(self.root) != self.rc0


class BlockingShadow:
    """shadow() of ONE Identity query as ONE blocking library call (psh_shadow_blocking): the fused launch reads the query
    from a pinned block and writes distances, indices AND the gathered paths into it, the call returns when the launch's
    completion words have landed there.  Results are handed to the caller as numpy arrays that VIEW the block (no copy out:
    11 us at k = 1024); a block is written again only once the caller has dropped every array of it -- callers that keep
    their results get a fresh block per call, up to POOL of them, after which results are copied out of a scratch block."""

    POOL = 8

    def __init__(self, rows: torch.Tensor, ds3: torch.Tensor, W: int, k: int, h: int, workspace: "Workspace", flags: int = 0):
        import numpy as np
        self._np = np
        self.rows = _dev_tensor(rows, torch.float32, "rows")
        self.ds3 = _dev_tensor(ds3, torch.float32, "dataset")
        self.W, self.k, self.h, self.flags = W, k, h, flags
        self.C = ds3.shape[1]
        L = load()
        lay = (C.c_size_t * 7)()
        _check(L.psh_shadow_block_layout(W, h, k, self.C, lay), "psh_shadow_block_layout")
        self.nbytes, self.o_status, self.o_query, self.o_hint, self.o_d, self.o_i, self.o_p = (int(v) for v in lay)
        R, T = self.rows.shape
        self.ws = workspace.get(workspace_bytes(R, T, 1, W, h, k))
        self._keep = workspace
        self._fn = L.psh_shadow_blocking
        self.blocks = [_HostBlock(self)]
        self.scratch = None
        self.last_fused = False

    def _block(self):
        for b in self.blocks:
            if not b.busy():
                return b, False
        if len(self.blocks) < self.POOL:
            b = _HostBlock(self)
            self.blocks.append(b)
            return b, False
        if self.scratch is None:
            self.scratch = _HostBlock(self)
        return self.scratch, True

    def call(self, stream_ptr: int, x, hint: float | None = None):
        """x: W float32 values (numpy).  Returns (status, None) or (0, (d (1,k), paths (1,k,C,W+h), idx (1,k,2)))."""
        np = self._np
        b, copy = self._block()
        b.q[:] = x
        a = b.args
        a[1] = stream_ptr
        if hint is not None:
            b.hint[0] = hint
            a[13] = 1
        else:
            a[13] = 0
        rc = self._fn(*a)
        if rc:
            _check(rc, "psh_shadow_blocking")
        self.last_fused = b.prof.path == 2
        st = int(b.status[0])
        if st != PSH_STATUS_OK:
            return st, None
        k, W, h, Cc = self.k, self.W, self.h, self.C
        d = np.ndarray((1, k), np.float32, b.root, self.o_d)
        idx = np.ndarray((1, k, 2), np.int32, b.root, self.o_i)
        paths = np.ndarray((1, k, Cc, W + h), np.float32, b.root, self.o_p)
        if copy:
            return 0, (d.copy(), paths.copy(), idx.copy())
        return 0, (d, paths, idx)


class PreparedStep:
    """One step of the row-sharded scan with everything but the stream and the query pointer fixed beforehand: the local
    psh_scan_topk into the send buffer, then psh_exchange_merge (all-gather + merge on the side stream).  The per-step host
    cost is two ctypes calls over pre-built argument lists -- no tensor is allocated, no Python-side check repeated (a rank
    that spends 100 us of Python per 85 us step is host-bound).  Buffers are owned here and REUSED by the next launch()."""

    def __init__(self, comm: "Comm", dataset: torch.Tensor, r_offset: int, B: int, W: int, k: int, h: int, workspace: "Workspace",
                 side: "torch.cuda.Stream", flags: int, sorted_merge: bool):
        ds = _dev_tensor(dataset, torch.float32, "dataset")
        dev = ds.device
        R, T = ds.shape
        G = comm.world
        self.B, self.W, self.k = B, W, k
        self.send = torch.empty(3 * B * k, dtype=torch.int32, device=dev)
        self.gathered = torch.empty((G, 3 * B * k), dtype=torch.int32, device=dev)
        self.out_d = torch.empty((B, k), dtype=torch.float32, device=dev)
        self.out_idx = torch.empty((B, k, 2), dtype=torch.int32, device=dev)
        # a status row PER LAUNCH (round-robin over STATUS_ROWS): a PSH_STATUS_RETRY of any step stays visible after later
        # launches of this slot have reused every other buffer (`status` is the row of the latest launch)
        self.status_all = torch.zeros((self.STATUS_ROWS, B), dtype=torch.int32, device=dev)
        self.status = self.status_all[0]
        self._launches = 0
        self.merge_ws = None
        if not sorted_merge:
            self.merge_ws = torch.empty(merge_workspace_bytes(B, k), dtype=torch.uint8, device=dev)
        self.ev_a, self.ev_b = torch.cuda.Event(), torch.cuda.Event()
        self.ev_a.record(); self.ev_b.record()                  # materialise the hipEvent handles
        ws = workspace.get(workspace_bytes(R, T, B, W, h, k))
        self._keep = (ds, ws, side, comm, workspace)
        self.prof = PshProfile()
        self.prof.mode = 1
        self.prof.flags = flags | (0 if sorted_merge else FLAG_UNSORTED)
        L = load()
        self._scan_fn, self._exch_fn = L.psh_scan_topk, L.psh_exchange_merge
        self._scan_args = [dev.index, None, ds.data_ptr(), R, T, r_offset, None, None, B, W, h, k, self.send.data_ptr(),
                           self.send.data_ptr() + 4 * B * k, self.status_all.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(self.prof)]
        self._status_base = self.status_all.data_ptr()
        self._exch_args = [comm._h, None, side.cuda_stream, self.send.data_ptr(), self.gathered.data_ptr(), B, k,
                           self.out_d.data_ptr(), self.out_idx.data_ptr(),
                           None if self.merge_ws is None else self.merge_ws.data_ptr(), 0 if self.merge_ws is None else self.merge_ws.numel(),
                           self.ev_a.cuda_event, self.ev_b.cuda_event]

    STATUS_ROWS = 1024

    def launch(self, stream_ptr: int, q_ptr: int) -> None:
        a = self._scan_args
        a[1] = stream_ptr
        a[6] = q_ptr
        row = self._launches % self.STATUS_ROWS
        self._launches += 1
        a[14] = self._status_base + 4 * self.B * row
        self.status = self.status_all[row]
        rc = self._scan_fn(*a)
        if rc:
            _check(rc, "psh_scan_topk")
        e = self._exch_args
        e[1] = stream_ptr
        rc = self._exch_fn(*e)
        if rc:
            _check(rc, "psh_exchange_merge")


def merge_topk(d_lists: torch.Tensor, idx_lists: torch.Tensor, k: int):
    """k best by (d, r, t) out of (B, n) candidates; entries with r < 0 are padding."""
    d = _dev_tensor(d_lists, torch.float32, "d_lists")
    ix = _dev_tensor(idx_lists, torch.int32, "idx_lists")
    B, n = d.shape
    if tuple(ix.shape) != (B, n, 2):
        raise ValueError("idx_lists must be (B, n, 2)")
    need = C.c_size_t(0)
    _check(load().psh_merge_workspace_bytes(B, k, C.byref(need)), "psh_merge_workspace_bytes")
    ws = torch.empty(int(need.value), dtype=torch.uint8, device=d.device)
    out_d = torch.empty((B, k), dtype=torch.float32, device=d.device)
    out_idx = torch.empty((B, k, 2), dtype=torch.int32, device=d.device)
    _check(load().psh_merge_topk(d.device.index, _stream_ptr(d.device), d.data_ptr(), ix.data_ptr(), B, n, k,
                                 out_d.data_ptr(), out_idx.data_ptr(), ws.data_ptr(), ws.numel()),
           "psh_merge_topk")
    return out_d, out_idx


def merge_topk_gathered(gathered: torch.Tensor, G: int, B: int, k_in: int, k: int):
    """Merge what one all-gather delivered.  `gathered`: int32 (G, 3*B*k_in): per rank the
    (B,k_in) float32 distances (bit pattern) followed by the (B,k_in,2) int32 indices."""
    g = _dev_tensor(gathered, torch.int32, "gathered")
    if tuple(g.shape) != (G, 3 * B * k_in):
        raise ValueError("gathered must be (G, 3*B*k_in) int32")
    need = C.c_size_t(0)
    _check(load().psh_merge_workspace_bytes(B, k, C.byref(need)), "psh_merge_workspace_bytes")
    ws = torch.empty(int(need.value), dtype=torch.uint8, device=g.device)
    out_d = torch.empty((B, k), dtype=torch.float32, device=g.device)
    out_idx = torch.empty((B, k, 2), dtype=torch.int32, device=g.device)
    base = g.data_ptr()
    # distances: rank stride 3*B*k_in floats; indices: the (r,t) pairs start B*k_in ints in
    # and their rank stride is 3*B*k_in/2 pairs -- whole (and 8-byte aligned) when B*k_in is even
    if (B * k_in) % 2:
        raise ValueError("B*k_in must be even for the in-place gathered merge")
    _check(load().psh_merge_topk_gathered(g.device.index, _stream_ptr(g.device), base, base + 4 * B * k_in,
                                          G, 3 * B * k_in, (3 * B * k_in) // 2, B, k_in, k,
                                          out_d.data_ptr(), out_idx.data_ptr(),
                                          ws.data_ptr(), ws.numel()), "psh_merge_topk_gathered")
    return out_d, out_idx


def merge_workspace_bytes(B: int, k: int) -> int:
    need = C.c_size_t(0)
    _check(load().psh_merge_workspace_bytes(B, k, C.byref(need)), "psh_merge_workspace_bytes")
    return int(need.value)


def merge_sorted_supported(G: int, k_in: int) -> bool:
    return G <= 64 and G * k_in <= 32768


def merge_sorted_gathered(gathered: torch.Tensor, G: int, B: int, k_in: int, k: int):
    """merge_topk_gathered for lists that are SORTED by (d, r, t), rank g owning smaller rows than rank g+1
    (psh_merge_sorted_gathered: positions by binary search, no selection, no sort)."""
    g = _dev_tensor(gathered, torch.int32, "gathered")
    if tuple(g.shape) != (G, 3 * B * k_in):
        raise ValueError("gathered must be (G, 3*B*k_in) int32")
    if (B * k_in) % 2:
        raise ValueError("B*k_in must be even for the in-place gathered merge")
    out_d = torch.empty((B, k), dtype=torch.float32, device=g.device)
    out_idx = torch.empty((B, k, 2), dtype=torch.int32, device=g.device)
    base = g.data_ptr()
    _check(load().psh_merge_sorted_gathered(g.device.index, _stream_ptr(g.device), base, base + 4 * B * k_in,
                                            G, 3 * B * k_in, (3 * B * k_in) // 2, B, k_in, k,
                                            out_d.data_ptr(), out_idx.data_ptr()), "psh_merge_sorted_gathered")
    return out_d, out_idx


def gather_paths(dataset: torch.Tensor, idx: torch.Tensor, length: int, r_offset: int = 0,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """out[b, i, c, :] = dataset[idx[b,i,0] - r_offset, c, idx[b,i,1] : +length]; dataset (R, C, T)."""
    ds = _dev_tensor(dataset, torch.float32, "dataset")
    ix = _dev_tensor(idx, torch.int32, "idx")
    if ds.dim() != 3:
        raise ValueError("dataset must be (R, C, T)")
    R, Cc, T = ds.shape
    n = ix.numel() // 2
    if out is None:
        out = torch.zeros(tuple(ix.shape[:-1]) + (Cc, length), dtype=torch.float32, device=ds.device)
    _check(load().psh_gather_paths(ds.device.index, _stream_ptr(ds.device), ds.data_ptr(), R, Cc, T, r_offset,
                                   ix.data_ptr(), n, length, out.data_ptr()), "psh_gather_paths")
    return out


def weighted_moments(values: torch.Tensor, weights: torch.Tensor | None):
    """(mean, std) over axis 1 of a (B, k, ...) float32 statistic with (B, k) float64 weights (None: uniform) --
    predict_from_paths()'s `proba.avg(values, axis=1)`, `proba.std(values, axis=1)` on the device; float64 (B, ...)."""
    v = _dev_tensor(values, torch.float32, "values")
    if v.dim() < 2:
        raise ValueError("values must be (B, k, ...)")
    B, k = v.shape[:2]
    m = v.numel() // (B * k) if B * k else 0
    if m == 0:
        raise ValueError("values is empty")
    w_ptr = None
    if weights is not None:
        w = _dev_tensor(weights, torch.float64, "weights")
        if tuple(w.shape) != (B, k):
            raise ValueError(f"weights must be (B, k) = ({B}, {k}), got {tuple(w.shape)}")
        w_ptr = w.data_ptr()
    mean = torch.empty((B,) + tuple(v.shape[2:]), dtype=torch.float64, device=v.device)
    std = torch.empty_like(mean)
    _check(load().psh_weighted_moments(v.device.index, _stream_ptr(v.device), v.data_ptr(), w_ptr, B, k, m,
                                       mean.data_ptr(), std.data_ptr()), "psh_weighted_moments")
    return mean, std


def _uniform_rows(x: torch.Tensor):
    """(n_rows, row_stride) when x (..., L) -- last dim contiguous -- is n_rows rows a constant stride apart (a contiguous
    tensor, or a slice of the last dimension of one: the out-context view of gathered paths), else None."""
    if x.dim() == 0 or x.stride(-1) != 1:
        return None
    if x.dim() == 1:
        return 1, x.shape[0]
    rs = x.stride(-2) if x.shape[-2] > 1 else None
    n = 1
    for dim in range(x.dim() - 2, -1, -1):
        if x.shape[dim] > 1:
            if rs is None:
                rs = x.stride(dim) // n if n else x.stride(dim)
            if x.stride(dim) != rs * n:
                return None
        n *= x.shape[dim]
    return n, (rs if rs is not None else x.shape[-1])


def realized_variance(x: torch.Tensor, Ts, vol: bool = False) -> torch.Tensor | None:
    """statistics.realized_variance on a float32 HIP tensor (..., L): (..., len(Ts)) float32, or None when this tensor's
    layout is not rows a constant stride apart (the caller then uses torch ops)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.numel() > 0):
        return None
    Ts = [int(T) for T in Ts]
    rows = _uniform_rows(x)
    if rows is None or not Ts or len(Ts) > 64 or min(Ts) <= 0:
        return None
    n_rows, row_stride = rows
    L = x.shape[-1]
    if row_stride < L:
        return None
    out = torch.empty(tuple(x.shape[:-1]) + (len(Ts),), dtype=torch.float32, device=x.device)
    arr = (C.c_int * len(Ts))(*Ts)
    _check(load().psh_realized_variance(x.device.index, _stream_ptr(x.device), x.data_ptr(), n_rows, row_stride, L, arr, len(Ts),
                                        1 if vol else 0, out.data_ptr()), "psh_realized_variance")
    return out


def count_nonfinite(x: torch.Tensor) -> int:
    """Number of NaN / +-inf samples of a float32 HIP tensor (one pass, one host synchronisation)."""
    t = _dev_tensor(x, torch.float32, "x")
    out = torch.empty((1,), dtype=torch.int64, device=t.device)
    _check(load().psh_count_nonfinite(t.device.index, _stream_ptr(t.device), t.data_ptr(), t.numel(), out.data_ptr()),
           "psh_count_nonfinite")
    return int(out.item())


def smear_nonfinite(dataset: torch.Tensor, back: int, fwd: int = 0) -> torch.Tensor:
    """(R, T) rows for the scan of an (R, C, T) ensemble that holds non-finite samples: NaN wherever any channel has one
    among the `fwd` samples before and the `back` samples after (the reference's zero-padded conv, see include/psh.h),
    channel 0 elsewhere."""
    ds = _dev_tensor(dataset, torch.float32, "dataset")
    if ds.dim() != 3:
        raise ValueError("dataset must be (R, C, T)")
    R, Cc, T = ds.shape
    out = torch.empty((R, T), dtype=torch.float32, device=ds.device)
    _check(load().psh_smear_nonfinite(ds.device.index, _stream_ptr(ds.device), ds.data_ptr(), R, Cc, T, int(back), int(fwd), out.data_ptr()),
           "psh_smear_nonfinite")
    return out


def rows_nonfinite(dataset: torch.Tensor) -> torch.Tensor:
    """(R,) int32 flags: 1 where a row of the (R, C, T) ensemble holds a NaN / +-inf sample in any channel."""
    ds = _dev_tensor(dataset, torch.float32, "dataset")
    if ds.dim() != 3:
        raise ValueError("dataset must be (R, C, T)")
    R, Cc, T = ds.shape
    out = torch.empty((R,), dtype=torch.int32, device=ds.device)
    _check(load().psh_rows_nonfinite(ds.device.index, _stream_ptr(ds.device), ds.data_ptr(), R, Cc, T, out.data_ptr()),
           "psh_rows_nonfinite")
    return out
