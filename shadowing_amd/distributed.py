"""R-sharded scan across the GPUs of one node (new work -- the
reference has no multi-GPU path).

Windows never cross trajectories, so the ensemble shards by rows: rank g keeps
rows [lo_g, hi_g) resident in its own HBM, runs the same scan with
`r_offset = lo_g`, and produces a local top-k of (d, r_global, t).  The only
exchange is ONE all-gather of B*k*12 bytes per rank (RCCL over xGMI when the
process group's backend is "nccl"); every rank then merges the G*k candidates
with the same (d, r, t) order, so the result is identical to the single-GPU and
to the reference CPU result for any G.  The payload is 12 KiB per query per rank:
latency-bound, nowhere near the xGMI link rate -- there is nothing to tune in the
collective, and there is no collective on the data path itself.
"""
from __future__ import annotations

import weakref
from typing import Callable

import numpy as np
import torch
import torch.distributed as dist

from . import _native
from .path_distance import RelativeMSE
from .path_embedding import Identity, PathEmbedding, PredictionContext


def _close_comm(comm, device) -> None:
    """Finalizer of a library-owned communicator (ShardedPathShadowing.close, object collection, interpreter exit)."""
    try:
        if torch.cuda.is_available():
            torch.cuda.synchronize(device)
        comm.close()
    except Exception:  # noqa: BLE001  (interpreter shutdown: the runtime may be gone already)
        pass


def shard_rows(R: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced row block of `rank`: [lo, hi)."""
    base, extra = divmod(R, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _native_local_topk(ds2d: torch.Tensor, q: torch.Tensor, k: int, h: int, r_offset: int, workspace,
                       out=None, check: bool = True, ker: torch.Tensor | None = None, unsorted: bool = False, flags: int = 0):
    """q: the query windows (B, W), or -- with `ker` (d, K), a linear embedding -- the embedded
    queries (B, d)."""
    def run(qq, exhaustive, out_):
        if ker is None:
            return _native.scan_topk(ds2d, qq, k, h=h, r_offset=r_offset, workspace=workspace, out=out_,
                                     exhaustive=exhaustive, unsorted=unsorted, flags=flags)
        return _native.scan_topk_embedded(ds2d, ker, qq, k, h=h, r_offset=r_offset, workspace=workspace, out=out_,
                                          exhaustive=exhaustive, flags=flags & _native.FLAG_EMBED_MX)
    if check and ker is None:    # one host sync: the status protocol (fused launch gave up -> separate launches; overflow -> exact)
        d, idx = _native.scan_topk_checked(ds2d, q, k, h=h, r_offset=r_offset, workspace=workspace, out=out, unsorted=unsorted,
                                           flags=flags)
        return d, idx, torch.zeros((q.shape[0],), dtype=torch.int32, device=q.device)
    d, idx, status = run(q, False, out)
    if check:    # one host sync: a query whose candidate slices overflowed is redone exactly
        bad = torch.nonzero(status != _native.PSH_STATUS_OK).flatten()
        if bad.numel():
            d2, idx2, _ = run(q[bad].contiguous(), True, None)
            d[bad] = d2
            idx[bad] = idx2
    return d, idx, status


class ShardedPathShadowing:
    """The (Identity | stock linear embedding) + RelativeMSE + PredictionContext scan over a
    row-sharded ensemble (BASELINE configs[3] and [4]).

    Every rank constructs it with ITS OWN rows (`local_dataset`, (R_local, 1, T) or
    (R_local, T)) and the global index of its first row.  `shadow()` is collective
    and returns the same global result on every rank.

    `local_topk` / `merge` are injection points for the CPU (gloo) tests of the
    exchange logic; production leaves them None and runs the HIP kernels.  With a linear
    embedding (Foveal, PathEmbedding(kernel)) `local_topk` receives the EMBEDDED queries.
    """

    def __init__(self, embedding: Identity, distance: RelativeMSE, local_dataset, row_offset: int,
                 context: PredictionContext | None = None, group=None, device: torch.device | None = None,
                 local_topk: Callable | None = None, merge: Callable | None = None, always_exchange: bool = False,
                 exchange: str = "auto", emulate_world: tuple | None = None, streams: int = 1, reserve_cus: bool = False):
        """`emulate_world = (G, n_windows_global, fill)`: this ONE process stands for rank 0 of a G-rank world whose
        collective is replaced by `fill(gathered, q, k)` -- it writes the lists of ranks 1..G-1 into rows 1..G-1 of the
        receive buffer (G, 3*B*k int32: (B,k) distance bits, then (B,k,2) indices), exactly where the all-gather would have
        left them.  Everything else (local scan into the send buffer, merge of the G lists) is the production code: how a
        full configs[3] (8 x 32768 rows) is exercised on one GPU.

        `streams` > 1 (HIP device; 2 is the measured optimum -- the exchange and the merge run on a stream of their own beside
        them: 88.7 us per step on one rank with the exchange forced, 105.5 with 3, 114.5 with 1): consecutive scan_begin() calls
        -- independent query batches -- issue their local scan round-robin on that many private streams, single queries as the overlap-friendly launches (PSH_FLAG_OVERLAP): the
        sample and the ranking of one step, and the exchange of another, run beside a third step's scan, and nothing in a
        scan waits for co-residency (no polling next to the collective's workgroups).  `reserve_cus=True` makes the private
        streams CU-masked ones (psh_stream_create_reserving: PSH_STREAM_RESERVED_CUS compute units stay free for the
        collective and the merge, whose workgroups otherwise wait for a scan block to leave); such streams are BLOCKING HIP
        streams (the masked-stream API takes no flags): call scan_begin() / finish() from a stream other than the legacy
        default stream then, or every step serialises with it.  Off by default: with the one-rank exchange that can be
        measured here it was slower (123 against 111 us per step), and no multi-GPU node was available to tune it on."""
        if type(distance) is not RelativeMSE:
            raise TypeError("the sharded scan implements RelativeMSE only")
        if type(embedding) is Identity:
            self._linear = False
        elif (isinstance(embedding, PathEmbedding) and type(embedding).forward is PathEmbedding.forward
              and embedding.kernel.ndim == 3 and embedding.kernel.shape[1] == 1
              and _native.embedding_supported(embedding.kernel.shape[0], embedding.kernel.shape[-1])):
            self._linear = True
        else:
            raise TypeError("the sharded scan implements Identity and stock linear embeddings (kernel (d,1,K) that "
                            "fits the native scan) only")
        self.embedding, self.distance = embedding, distance
        self._ker = None
        self.context = context or PredictionContext(horizon=None)
        if type(self.context) is not PredictionContext:
            raise TypeError("the sharded scan implements PredictionContext only")
        self.group = group
        self.always_exchange = always_exchange      # run the all-gather + merge even with one rank (tests)
        # who runs the collective: "library" = libpsh_hip.so itself (psh_exchange_merge: RCCL all-gather + merge on a
        # side stream behind one event, one C call per step); "torch" = torch.distributed's all_gather_into_tensor
        # (asynchronous) and the merge on the compute stream; "auto" = the library on a HIP device, torch otherwise
        if exchange not in ("auto", "library", "torch"):
            raise ValueError('exchange must be "auto", "library" or "torch"')
        self.exchange = exchange
        self.fuse = True            # False: the local scan as separate launches (PSH_FLAG_NO_FUSE)
        self._emulate = emulate_world
        if emulate_world is not None:
            self.always_exchange, self.exchange = True, "torch"
        self._comm = None
        self._side = None
        self._events = None
        self._ev_next = 0
        self.row_offset = int(row_offset)
        self._local_topk = local_topk
        self._merge = merge
        self.last_status = None
        ds = local_dataset if isinstance(local_dataset, torch.Tensor) else torch.as_tensor(
            np.ascontiguousarray(local_dataset), dtype=torch.float32)
        if ds.dim() == 2:
            ds = ds[:, None, :]
        if ds.dim() != 3 or ds.shape[1] != 1:
            raise ValueError("local_dataset must be (R_local, 1, T) or (R_local, T)")
        self._n_global = None       # windows of the whole ensemble (summed over the ranks on first use)
        if local_topk is None:   # production: resident in this rank's HBM
            if device is None:
                if not torch.cuda.is_available():
                    raise _native.NativeLibraryError("ShardedPathShadowing needs a HIP device")
                device = torch.device("cuda", torch.cuda.current_device())
            _native.load()
            ds = ds.to(device)
            self._workspace = _native.Workspace(device)
            if self._linear:
                self._ker = embedding.kernel[:, 0, :].to(device=device, dtype=torch.float32).contiguous()
                from .path_embedding import Foveal
                self._emb_flags = 0 if isinstance(embedding, Foveal) else _native.FLAG_EMBED_MX
        else:
            self._workspace = None
        self.dataset = ds.contiguous()
        self.device = self.dataset.device
        # the rows the native scan reads: the shard itself, or -- it holds NaN / +-inf samples -- a copy with them written back
        # over the horizon, so that windows are NaN exactly where the reference's zero-padded conv makes them NaN
        # (PathShadowing._scan_rows_of, psh_prep.hip); paths are gathered from `dataset`
        self._rows = self.dataset[:, 0, :]
        dirty = bool(local_topk is None and self.dataset.numel() and _native.count_nonfinite(self.dataset))
        self._dirty_split = None
        if dirty and self._linear:
            # NaN / +-inf samples behind a linear embedding (round 6: served, no longer refused): the embedded scans' rejection
            # tests assume finite data (prefix sums and matrix-core tiles spread a NaN over clean windows), so THIS rank scans its
            # clean rows as ever and its dirty rows -- the horizon smeared in -- with the exhaustive dense chains, which meet a NaN
            # exactly where the reference's zero-padded conv does (ref path_embedding.py:48-51, :129-132), and merges the two
            # lists before the exchange: PathShadowing._split_dirty_rows per shard.  No rank has to know about another's shard.
            back = int(self.context.get_out_times())
            flags_ = _native.rows_nonfinite(self.dataset)
            dirty_idx = torch.nonzero(flags_).flatten()
            clean_idx = torch.nonzero(flags_ == 0).flatten()
            clean_rows = self.dataset[clean_idx, 0, :].contiguous()
            dirty_rows = _native.smear_nonfinite(self.dataset[dirty_idx].contiguous(), back, 0)
            self._dirty_split = (clean_idx, clean_rows, dirty_idx, dirty_rows)
            dirty = False                               # (self._rows is not used by the split scan)
        if dirty:
            self._rows = _native.smear_nonfinite(self.dataset, int(self.context.get_out_times()), 0)
        self._scan_streams = None
        self._reserve = 0
        if streams > 1 and local_topk is None and self.device.type == "cuda":
            # the scans of consecutive steps overlap on these streams and would take EVERY compute unit between them: the
            # collective and the merge on the side stream get theirs through a CU mask (psh_stream_create_reserving)
            made = ([_native.reserving_stream(self.device) for _ in range(int(streams))] if reserve_cus
                    else [(torch.cuda.Stream(self.device), 0) for _ in range(int(streams))])
            self._scan_streams = [m[0] for m in made]
            self._reserve = min(m[1] for m in made)
            self._scan_ws = [_native.Workspace(self.device) for _ in range(int(streams))]
        self._step = 0
        self._fast = {}             # (B, W, k) -> ring of _native.PreparedStep (private streams, library exchange)

    @property
    def world_size(self) -> int:
        if self._emulate is not None:
            return int(self._emulate[0])
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def n_windows_global(self) -> int:
        """Admissible windows of the whole (sharded) ensemble: one all-reduce, once."""
        if self._n_global is None and self._emulate is not None:
            self._n_global = int(self._emulate[1])
        if self._n_global is None:
            h = self.context.get_out_times()
            R_local, _, T = self.dataset.shape
            n = R_local * max(T - self.embedding.kernel.shape[-1] - h + 1, 0)
            if self.world_size > 1:
                t = torch.tensor([n], dtype=torch.int64, device=self.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                n = int(t.item())
            self._n_global = n
        return self._n_global

    def _library_exchange(self):
        """The communicator owned by libpsh_hip.so, created collectively on first use (the 128-byte id travels through
        the torch.distributed group that is there anyway), its side stream and a ring of pre-recorded events (an event
        record is a packet on the stream: none is spent per step beyond the two the hand-over needs)."""
        if self._comm is None:
            G = self.world_size
            rank = dist.get_rank(self.group) if dist.is_initialized() else 0
            err = None
            box = [None]
            if rank == 0:
                try:
                    box[0] = _native.comm_unique_id()
                except Exception as e:                               # noqa: BLE001 -- (RCCL could not be opened: the ranks hear of it below)
                    err = e
            if G > 1:
                dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                           group=self.group)
            # Created collectively, and agreed on collectively: a rank whose library cannot open RCCL or whose ncclCommInitRank
            # fails (first contact with a node's links) must not leave the others inside the next all-gather.  Every rank reports
            # through the torch.distributed group that is there anyway; one failure -> every rank drops its communicator and the
            # object serves the exchange with torch.distributed.all_gather (exchange="torch") from here on, with ONE warning.
            if box[0] is None:
                err = err or _native.NativeLibraryError("rank 0 could not make an RCCL unique id")
            else:
                try:
                    self._comm = _native.Comm(self.device, G, rank, box[0])
                except Exception as e:                               # noqa: BLE001 -- whatever the library raised: agree first
                    err = e
            if G > 1:
                flag = torch.tensor([0 if err is None else 1], dtype=torch.int32,
                                    device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
                failed = bool(int(flag.item()))
            else:
                failed = err is not None
            if failed:
                if self._comm is not None:
                    _close_comm(self._comm, self.device)
                    self._comm = None
                if self.exchange == "library":
                    raise _native.NativeLibraryError(f'exchange="library": psh_comm_create failed on a rank ({err!r})')
                import warnings
                warnings.warn("ShardedPathShadowing: the library's RCCL communicator could not be created on every rank "
                              f"({err!r} on this one); the exchange runs through torch.distributed.all_gather instead",
                              RuntimeWarning, stacklevel=3)
                self.exchange = "torch"
                return None
            # the communicator belongs to libpsh_hip.so: it goes away with this object (or at interpreter exit)
            self._comm_finalizer = weakref.finalize(self, _close_comm, self._comm, self.device)
            # not at interpreter exit: a synchronize / ncclCommDestroy there hangs when a peer rank is already gone with a
            # collective pending; close() is the orderly path (the OS reclaims the rest)
            self._comm_finalizer.atexit = False
            self._side = torch.cuda.Stream(device=self.device)
            self._events = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(8)]
            for a, b in self._events:              # materialise the hipEvent handles
                a.record(); b.record()
        return self._comm

    def _use_library(self, B: int, k: int) -> bool:
        usable = (self._local_topk is None and self._merge is None and self.device.type == "cuda" and (B * k) % 2 == 0)
        if self.exchange == "library" and not usable:
            raise _native.NativeLibraryError(
                'exchange="library" (psh_exchange_merge) needs a HIP device, the native scan and an even B*k '
                f'(device {self.device}, B*k = {B * k}); use exchange="auto" or "torch"')
        return usable and self.exchange != "torch"

    def status_max(self) -> int:
        """Largest status word of ANY step issued through the prepared ring since reset_status() (one host
        synchronisation): 0 = every step's results were valid; PSH_STATUS_RETRY = some step gave up.  `last_status` only
        shows the latest launch of one slot."""
        worst = 0
        for ring in self._fast.values():
            for slot in ring:
                worst = max(worst, int(slot.status_all.max().item()))
        return worst

    def reset_status(self) -> None:
        """Forget the status history (call with the device idle, e.g. after a synchronize)."""
        for ring in self._fast.values():
            for slot in ring:
                slot.status_all.zero_()

    def close(self):
        if self._comm is not None:
            self._comm_finalizer.detach()
            _close_comm(self._comm, self.device)
            self._comm = None

    def local_scan(self, q: torch.Tensor, k: int, out=None, check: bool = True, unsorted: bool = False, flags: int = 0,
                   workspace=None):
        """This rank's candidates: (d (B,k), idx (B,k,2), status) with global row numbers,
        padded with (+inf, -1) when the shard holds fewer than k windows."""
        h = self.context.get_out_times()
        R_local, _, T = self.dataset.shape
        n_local = R_local * max(T - self.embedding.kernel.shape[-1] - h + 1, 0)
        k_local = min(k, n_local)
        if k_local == 0:                            # an empty shard (fewer rows than ranks): pure padding
            B = q.shape[0]
            d = torch.full((B, k), float("inf"), dtype=torch.float32, device=self.device)
            idx = torch.full((B, k, 2), -1, dtype=torch.int32, device=self.device)
            if out is not None:
                out[0].copy_(d)
                out[1].copy_(idx)
                d, idx = out
            return d, idx, torch.zeros((B,), dtype=torch.int32, device=self.device)
        if self._local_topk is not None:            # CPU test path (oracle injected)
            d, idx = self._local_topk(self.dataset[:, 0, :], q, k_local, h, self.row_offset)
            status = None
        elif self._dirty_split is not None:
            d, idx = self._split_local_topk(q, k_local, h, workspace or self._workspace, flags)
            status = torch.zeros((q.shape[0],), dtype=torch.int32, device=self.device)
            if out is not None and k_local == k:
                out[0].copy_(d)
                out[1].copy_(idx)
                d, idx = out
        else:
            d, idx, status = _native_local_topk(self._rows, q, k_local, h, self.row_offset, workspace or self._workspace,
                                                out=out if k_local == k else None, check=check, ker=self._ker,
                                                unsorted=unsorted, flags=flags)
        if k_local < k:
            B = q.shape[0]
            d = torch.cat([d, d.new_full((B, k - k_local), float("inf"))], dim=1)
            idx = torch.cat([idx, idx.new_full((B, k - k_local, 2), -1)], dim=1)
        return d.contiguous(), idx.contiguous(), status

    def _split_local_topk(self, hx: torch.Tensor, k_local: int, h: int, workspace, flags: int):
        """This rank's k_local best behind a linear embedding when its shard holds non-finite samples: the clean rows through the
        embedded scan (status protocol: a query whose slices overflow is redone exactly), the dirty rows -- smeared -- through the
        exhaustive dense chains, the two lists merged by (d, r, t); row numbers global.  One host synchronisation (rare path)."""
        clean_idx, clean_rows, dirty_idx, dirty_rows = self._dirty_split
        Tp = self.dataset.shape[-1] - self._ker.shape[-1] - h + 1
        n_clean, n_dirty = int(clean_idx.numel()) * Tp, int(dirty_idx.numel()) * Tp
        parts_d, parts_i = [], []
        if n_clean > 0:
            kc = min(k_local, n_clean)
            dc, ic, _ = _native_local_topk(clean_rows, hx, kc, h, 0, workspace, check=True, ker=self._ker, flags=flags)
            ic = ic.clone()
            ic[..., 0] = (clean_idx[ic[..., 0].long()] + self.row_offset).to(torch.int32)
            parts_d.append(dc)
            parts_i.append(ic)
        if n_dirty > 0:
            kd = min(k_local, n_dirty)
            dd, idd, _ = _native.scan_topk_embedded(dirty_rows, self._ker, hx, kd, h=h, workspace=workspace, exhaustive=True,
                                                    flags=_native.FLAG_EMBED_DENSE)
            idd = idd.clone()
            idd[..., 0] = (dirty_idx[idd[..., 0].long()] + self.row_offset).to(torch.int32)
            parts_d.append(dd)
            parts_i.append(idd)
        if len(parts_d) == 1 and parts_d[0].shape[1] == k_local:
            return parts_d[0].contiguous(), parts_i[0].contiguous()
        return _native.merge_topk(torch.cat(parts_d, dim=1).contiguous(), torch.cat(parts_i, dim=1).contiguous(), k_local)

    def scan(self, queries: torch.Tensor, k: int, check: bool = True):
        """Collective.  queries (B, W) float32 (same on every rank).  Returns device
        tensors (d (B,k), idx (B,k,2)) -- the global k best, identical on all ranks.

        One exchange: every rank's scan writes (d | idx) straight into its send buffer
        (3*B*k int32), ONE all-gather moves it, and the merge kernel reads the G lists
        where the collective left them (no pack / unpack copies).  `check=False` skips the
        per-call host synchronisation that looks at the overflow status (benchmark loops
        check `last_status` once at the end)."""
        return self.scan_begin(queries, k, check=check).finish()

    def scan_begin(self, queries: torch.Tensor, k: int, check: bool = True, queries_ready: bool = False) -> "PendingScan":
        """`queries_ready` (private streams only): the caller's word that `queries` is already materialised in HBM (nothing
        still enqueued on the current stream writes it): the step's private stream then does not wait for the current
        stream -- which, in a pipelined loop, has just been told to wait for the merged result of an EARLIER step, a
        dependency the queries do not have."""
        if self._scan_streams is None:
            return self._scan_begin(queries, k, check, None)
        fast = self._fast_step(queries, k, check, queries_ready)
        if fast is not None:
            return fast
        # this step's stream: it sees everything the caller's stream has enqueued so far (the queries), and the caller's
        # stream sees the results through PendingScan.finish()
        i = self._step % len(self._scan_streams)
        self._step += 1
        s = self._scan_streams[i]
        if not queries_ready:
            s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            return self._scan_begin(queries, k, check, self._scan_ws[i])

    def _fast_step(self, queries, k: int, check: bool, queries_ready: bool = False):
        """The lean form of a step on the private streams (a host that spends more Python per step than the GPU spends
        scanning is the bottleneck): Identity scan, library exchange, queries already a float32 (B, W) tensor on the device,
        no per-call status check.  Prepared argument lists and a ring of buffers per (B, W, k) (_native.PreparedStep); two
        ctypes calls per step.  Results live in the ring: valid until 2 x streams further scan_begin() calls -- the GPU
        orders a slot's next scan and exchange behind its previous exchange, NOT behind whoever reads the merged result:
        a consumer enqueues its reads of (d, idx) before it begins the 2 x streams-th step after this one (a pipelined
        loop that finishes step i right after beginning step i + 1 does)."""
        if (check or self._linear or self._local_topk is not None or self._merge is not None or self._emulate is not None
                or not isinstance(queries, torch.Tensor) or queries.device != self.device or queries.dtype != torch.float32
                or queries.dim() != 2 or not queries.is_contiguous() or not self.fuse):
            return None
        B, W = queries.shape
        G = self.world_size
        if not ((G > 1 or self.always_exchange) and self._use_library(B, k)):
            return None
        h = self.context.get_out_times()
        R_local, _, T = self.dataset.shape
        if W != self.embedding.kernel.shape[-1] or R_local * max(T - W - h + 1, 0) < k or k > self.n_windows_global():
            return None                                      # (short shards, errors: the general path pads / raises)
        key = (B, W, k)
        ring = self._fast.get(key)
        if ring is None:
            comm = self._library_exchange()
            if comm is None:                                 # (no communicator on some rank: the general path, torch's all-gather)
                return None
            n = len(self._scan_streams)
            sorted_merge = _native.merge_sorted_supported(G, k)
            ring = [_native.PreparedStep(comm, self._rows, self.row_offset, B, W, k, h, self._scan_ws[j % n], self._side,
                                         (_native.FLAG_OVERLAP | (_native.FLAG_RESERVE_CUS if self._reserve else 0)) if B == 1 else 0,
                                         sorted_merge) for j in range(2 * n)]
            torch.cuda.synchronize(self.device)
            self._fast[key] = ring
        j = self._step % len(ring)
        self._step += 1
        slot = ring[j]
        s = self._scan_streams[j % len(self._scan_streams)]
        if not queries_ready:
            s.wait_stream(torch.cuda.current_stream(self.device))
        # the slot's buffers are reused: this scan writes `send`, which the all-gather of the slot's PREVIOUS step reads on
        # the side stream, and this step's merge overwrites `gathered` / `out_*` behind it -- so the scan waits for that
        # exchange's end (ev_b still carries its record; 2 x streams steps ago, long complete unless a rank lags in the
        # collective -- exactly when the order matters)
        s.wait_event(slot.ev_b)
        queries.record_stream(s)
        slot.launch(s.cuda_stream, queries.data_ptr())
        self.last_status = slot.status
        return PendingScan(self, None, (None, None, "library", slot.ev_b, "ring"), (slot.out_d, slot.out_idx), B, k)

    def _scan_begin(self, queries: torch.Tensor, k: int, check: bool, ws) -> "PendingScan":
        """First half of scan(): the local scan and the START of the all-gather (async_op).  `finish()` on
        the returned handle makes the compute stream wait for the collective and merges.  A stream of
        independent query batches (rolling query dates) is pipelined by beginning batch i+1 before finishing
        batch i: the all-gather of i (its own RCCL stream, ~20 us of latency for 12 KiB per rank) then runs under
        the scan of i+1 instead of in front of the merge.  Every handle owns its send / receive buffers."""
        if self._linear:
            # the query embedding is the module's own conv1d (reference path_shadowing.py:140), on
            # the module's device; what travels to the scan is (B, d)
            from .path_shadowing import single_thread
            kdev = self.embedding.kernel.device
            with single_thread():
                queries = self.embedding(queries.to(kdev, dtype=torch.float32)[:, None, :])[:, 0, :]
        q = queries.to(self.device, dtype=torch.float32).contiguous()
        if ws is not None and q.device.type == "cuda":
            q.record_stream(torch.cuda.current_stream(self.device))   # read on this private stream, maybe allocated on the caller's
        B = q.shape[0]
        G = self.world_size
        if k > self.n_windows_global():
            # the reference fails inside torch.topk with this exception type (path_shadowing.py:165)
            raise RuntimeError("selected index k out of range")
        exchange = G > 1 or self.always_exchange
        native = self._local_topk is None and self._merge is None
        if native and (B * k) % 2 == 0:
            send = torch.empty(3 * B * k, dtype=torch.int32, device=self.device)
            out = (send[:B * k].view(torch.float32).view(B, k), send[B * k:].view(B, k, 2))
            # merging SORTED per-rank lists is a binary-search count per entry (psh_merge_sorted_gathered: no
            # selection, no sort, 15 us at G = 8 against 36 us for the general merge); lists too long for
            # its LDS go to the general merge, which orders anyway -- the local selection then skips its own
            sorted_merge = _native.merge_sorted_supported(G, k)
            library = exchange and self._use_library(B, k)
            comm = self._library_exchange() if library else None
            if library and comm is None:                     # (no communicator on some rank: torch's all-gather, every rank alike)
                library = False
            # private scan streams: nothing in the scan needs co-residency, so no compute unit is reserved for the collective
            mode = ((_native.FLAG_OVERLAP | (_native.FLAG_RESERVE_CUS if self._reserve else 0)) if ws is not None
                    else (_native.FLAG_RESERVE_CUS if library else 0))
            d, idx, self.last_status = self.local_scan(q, k, out=out, check=check, unsorted=exchange and not sorted_merge,
                                                       flags=(mode if self.fuse else _native.FLAG_NO_FUSE) | getattr(self, "_emb_flags", 0),
                                                       workspace=ws)
            if d.data_ptr() != out[0].data_ptr():     # shard smaller than k: padded copies were made
                out[0].copy_(d)
                out[1].copy_(idx)
            if not exchange:
                return PendingScan(self, None, None, (d, idx), B, k)
            if library:
                # ONE C call: record an event on this stream, and on the side stream behind it the RCCL all-gather and
                # the merge; the next batch's scan starts on this stream right away
                gathered = torch.empty((G, 3 * B * k), dtype=torch.int32, device=self.device)
                out_d = torch.empty((B, k), dtype=torch.float32, device=self.device)
                out_idx = torch.empty((B, k, 2), dtype=torch.int32, device=self.device)
                merge_ws = None
                if not sorted_merge:
                    merge_ws = torch.empty(_native.merge_workspace_bytes(B, k), dtype=torch.uint8, device=self.device)
                for t in (send, gathered, out_d, out_idx) + ((merge_ws,) if merge_ws is not None else ()):
                    t.record_stream(self._side)        # allocated on this stream, used on the side stream
                ev_a, ev_b = self._events[self._ev_next % len(self._events)]
                self._ev_next += 1
                comm.exchange_merge(send, gathered, B, k, out_d, out_idx, merge_ws, self._side, ev_a, ev_b)
                return PendingScan(self, None, (send, gathered, "library", ev_b, merge_ws), (out_d, out_idx), B, k)
            gathered = torch.empty((G, 3 * B * k), dtype=torch.int32, device=self.device)
            if self._emulate is not None:
                gathered[0].copy_(send)
                self._emulate[2](gathered, q, k)
                return PendingScan(self, _Done(), (send, gathered, "sorted" if sorted_merge else "general"), None, B, k)
            work = dist.all_gather_into_tensor(gathered.view(-1), send, group=self.group, async_op=True)
            return PendingScan(self, work, (send, gathered, "sorted" if sorted_merge else "general"), None, B, k)
        d, idx, self.last_status = self.local_scan(q, k, check=check, flags=getattr(self, "_emb_flags", 0), workspace=ws)
        if not exchange:
            return PendingScan(self, None, None, (d, idx), B, k)
        # generic form (CPU tests, odd B*k): pack (d, r, t) as 3 x int32, one all-gather
        packed = torch.cat([d.view(torch.int32).unsqueeze(-1), idx], dim=-1).contiguous()   # (B, k, 3)
        gathered = torch.empty((G * B, k, 3), dtype=torch.int32, device=self.device)   # rank-major concat
        work = dist.all_gather_into_tensor(gathered, packed, group=self.group, async_op=True)
        return PendingScan(self, work, (packed, gathered, "packed"), None, B, k)

    def shadow(self, x_context, k: int = 1):
        """Collective counterpart of PathShadowing.shadow(): numpy (d (B,k),
        paths (B,k,1,W+h), idx (B,k,2)), the same on every rank.  Paths are collected
        from their owner ranks with one sum all-reduce (each entry has exactly one
        non-zero contributor, so the sum is exact)."""
        x = x_context if isinstance(x_context, torch.Tensor) else torch.as_tensor(np.asarray(x_context), dtype=torch.float32)
        x = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
        if x.shape[-1] != self.embedding.kernel.shape[-1]:
            raise Exception("The embedding kernel should be of the same size as the context.")
        d, idx = self.scan(x, k)
        length = x.shape[-1] + self.context.get_out_times()
        if self._local_topk is None:
            paths = _native.gather_paths(self.dataset, idx, length, r_offset=self.row_offset)
        else:   # CPU test path of the exchange logic
            paths = torch.zeros(tuple(idx.shape[:-1]) + (1, length), dtype=torch.float32, device=self.device)
            r = idx[..., 0].long() - self.row_offset
            mine = (r >= 0) & (r < self.dataset.shape[0])
            t = idx[..., 1].long()
            for b, i in zip(*torch.nonzero(mine, as_tuple=True)):
                paths[b, i, 0] = self.dataset[r[b, i], 0, t[b, i]:t[b, i] + length]
        if self.world_size > 1 and self._emulate is None:
            dist.all_reduce(paths, op=dist.ReduceOp.SUM, group=self.group)
        return d.cpu().numpy(), paths.cpu().numpy(), idx.cpu().numpy()


class _Done:
    """The 'collective' of an emulated world: nothing to wait for."""
    def wait(self):
        return None


class PendingScan:
    """A sharded scan whose all-gather is in flight (ShardedPathShadowing.scan_begin)."""

    def __init__(self, owner: ShardedPathShadowing, work, buffers, local, B: int, k: int):
        self._owner, self._work, self._buffers, self._result, self._B, self._k = owner, work, buffers, local, B, k
        # the private stream this step was begun on, if any (scan_begin runs under `with torch.cuda.stream(s)`)
        cur = torch.cuda.current_stream(owner.device) if owner.device.type == "cuda" else None
        self._stream = cur if (owner._scan_streams is not None and cur in owner._scan_streams) else None

    def finish(self):
        """(d (B,k), idx (B,k,2)) on the device, identical on all ranks.  Waits for the collective on the
        compute STREAM (no host synchronisation with the RCCL backend) and merges the gathered lists."""
        cur = torch.cuda.current_stream(self._owner.device) if self._owner.device.type == "cuda" else None
        if self._buffers is not None and self._buffers[2] == "library":
            # the merged lists are written by the side stream: whoever consumes them on this stream waits for the event
            cur.wait_event(self._buffers[3])
            ring = self._buffers[4] == "ring"                # (the lean path's results live in buffers the owner keeps)
            self._buffers = None
            if not ring:
                for t in self._result:
                    t.record_stream(cur)
            return self._result
        if self._stream is not None:
            # begun on one of the owner's private streams: what follows here (the wait for the collective, the merge, the
            # consumer) is on the caller's stream, behind everything that private stream has enqueued for this step
            cur.wait_stream(self._stream)
            self._stream = None
            if self._result is not None:
                for t in self._result:
                    t.record_stream(cur)
            if self._buffers is not None:
                for t in self._buffers[:2]:                  # send / gathered: allocated on the private stream, merged on this one
                    if isinstance(t, torch.Tensor):
                        t.record_stream(cur)
        if self._result is None:
            o, B, k = self._owner, self._B, self._k
            G = o.world_size
            self._work.wait()
            _, gathered, kind = self._buffers
            if kind == "sorted":
                self._result = _native.merge_sorted_gathered(gathered, G, B, k, k)
            elif kind == "general":
                self._result = _native.merge_topk_gathered(gathered, G, B, k, k)
            else:
                allc = gathered.view(G, B, k, 3).permute(1, 0, 2, 3).reshape(B, G * k, 3)
                d_all = allc[..., 0].contiguous().view(torch.float32)
                i_all = allc[..., 1:].contiguous()
                self._result = (o._merge or _native.merge_topk)(d_all, i_all, k)
            self._work = self._buffers = None
        return self._result
