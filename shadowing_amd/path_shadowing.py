"""PathShadowing: the k-nearest-path scan, MI355X-native behind the reference's API.

Mirror of RudyMorel/shadowing shadowing/path_shadowing/path_shadowing.py
(`PathShadowing` :61-301, helpers :16-58).  The object, its methods, their
arguments, return types and error behaviour are the reference's; what happens
inside `batched_distance(..., cuda=True)` is new:

  * Identity embedding + RelativeMSE distance + PredictionContext, single channel
    -> the hand-written HIP kernels of libpsh_hip.so (include/psh.h), with the
    trajectory ensemble RESIDENT in HBM (uploaded once per dataset, not once per
    split per call as path_shadowing.py:154-155 does), exact fp32 distances in
    the reference CPU path's arithmetic order, rows ordered by (d, r, t).
  * any other linear embedding whose forward is the stock conv1d (Foveal,
    PathEmbedding(kernel)) + RelativeMSE + PredictionContext -> psh_scan_topk_embedded:
    embedding, distance, filter and top-k fused in one pass over the resident ensemble.
  * any other plugin combination (user subclasses overriding forward, other
    contexts, several channels) -> a generic torch path with the reference's
    semantics, on the HIP device when cuda=True.

`cuda=True` never falls back to the CPU: if the HIP library is missing or no
device is present it raises.  `cuda=False` is the reference's own meaning: run
the generic torch formulation on the host.
"""
from __future__ import annotations

import contextlib
import math
import warnings
import weakref
from pathlib import Path
from typing import Callable

import numpy as np
import torch
from tqdm import tqdm

from . import _native
from .averaging import HAVE_SCATSPECTRA, DiscreteProba, Softmax, Uniform
from .path_distance import PathDistance, RelativeMSE
from .path_embedding import (ArrayType, ContextManagerBase, CrossChannelContext, Foveal, Identity, ImputationContext,
                             PathEmbedding, PredictionContext)


@contextlib.contextmanager
def single_thread():
    """The (tiny) query embedding on the host with ONE intra-op thread: the result is the same bits (checked
    on every fixture), but an 8..256-thread OpenMP team woken for 34 x 126 multiply-adds stalls for up to
    200 ms every few dozen calls (measured: p99 160 ms against a 0.17 ms median; 0.8 ms worst case with one)."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        yield
    finally:
        torch.set_num_threads(n)


def _dim_array(x: ArrayType) -> ArrayType:
    """(T,) -> (1,1,T);  (B,T) -> (B,1,T);  (B,C,T) unchanged  (ref :16-26)."""
    if x is None:
        return x
    if x.ndim == 1:
        return x[None, None, :]
    if x.ndim == 2:
        return x[:, None, :]
    if x.ndim == 3:
        return x
    raise Exception("Array cannot be formatted to (B, C, T) shape.")


def _torch(x: ArrayType) -> torch.Tensor:
    """float32 torch tensor (a tensor is returned as is)  (ref :29-33)."""
    return x if isinstance(x, torch.Tensor) else torch.tensor(x, dtype=torch.float32)


def _numpy(x: ArrayType) -> np.ndarray:
    return x if isinstance(x, np.ndarray) else x.cpu().numpy()


def select_cartesian_product(indices: torch.Tensor, tensors: list[torch.Tensor]) -> torch.Tensor:
    """Rows `indices` of torch.cartesian_prod(*tensors) without building the product
    (ref :43-58): mixed-radix decode of the flat index, most significant factor first."""
    sizes = [int(t.shape[0]) for t in tensors]
    columns = []
    rest = indices
    weight = 1
    for s in sizes:
        weight *= s
    for t, s in zip(tensors, sizes):
        weight //= s
        columns.append(t[(rest // weight) % s])
    return torch.stack(columns, dim=-1)


class PathShadowing:
    """Scan a dataset of generated paths for the ones closest to an observed context.

    Attributes (as in the reference): `embedding`, `distance`, `dataset`, `context`.
    """

    def __init__(self, embedding: PathEmbedding, distance: PathDistance,
                 dataset, context: ContextManagerBase | None = None, cache: bool | str = "auto", hint: str | None = None):
        """`cache` governs the copy of the ensemble kept in HBM for cuda=True (the reference re-reads and
        re-uploads `dataset` on every call, ref :205, :154-155):
          "auto" (default)  kept only when staleness is detectable or impossible: a torch tensor (CUDA: used in
                            place; CPU: keyed on its version counter, which in-place torch ops bump) or a
                            read-only numpy array; a WRITEABLE numpy array is uploaded on every call, exactly
                            as the reference does, so an in-place edit is never missed;
          True              kept for any dataset until `refresh()` -- the caller promises to call it after
                            editing the array in place (8.5 ms per call saved at R = 32768 x T = 4096);
          False             never kept.
        `hint` (cuda=True, one Identity query per call -- the blocking shadow() of a loop over query dates):
          None (default)    every call samples the ensemble for its admission level;
          "auto"            a call hands the library the level the PREVIOUS call's k-th distance predicts for this query
                            (psh_profile.tau_hint = (d_k ||x||)^2 x 1.15): the fused launch then runs no sample phase and no first
                            grid barrier.  A hint that falls short (fewer than k windows below it, or more than the candidate
                            lists hold) costs one more launch without it and switches the hints off for a few calls
                            (`last_hint` says what happened) -- results are the exact top-k either way.  Pays for rolling query
                            dates whose k-th distance moves by a few per cent from date to date; not for unrelated queries."""
        if isinstance(dataset, Path) or hasattr(dataset, "load"):
            dataset = self._load_with_scatspectra(dataset)
        if cache not in (True, False, "auto"):
            raise ValueError('cache must be True, False or "auto"')
        if hint not in (None, "auto"):
            raise ValueError('hint must be None or "auto"')
        self.hint = hint
        self._hint_state = None     # {"dk": relative k-th distance of the last call, "k", "W", "skip": calls left without a hint, "fails"}
        self.last_hint = None       # None: the last call gave no hint; "ok" / "short": what became of the one it gave
        self._warned_reupload = False
        self._predict_scope = None  # (dataset object, device copy) while predict() loops over its context splits
        self.dataset = dataset
        self.embedding = embedding
        self.distance = distance
        self.context = context or PredictionContext(horizon=None)
        self.cache = cache
        self._resident = None       # (key, device tensor (R, C, T), weakref to the host tensor) -- the ensemble in HBM
        self._scan_rows = None      # (key, device tensor (R, T)): channel 0 of a multi-channel ensemble
        self._dirty = False         # the resident ensemble holds NaN / +-inf samples (set by _scan_rows_of)
        self._dirty_split = None    # (back, clean rows' indices, their rows, dirty rows' indices, their smeared rows): _split_dirty_rows
        self._served_by = "hip"     # what the last _native_scan ran: "hip", or "torch" (a dirty ensemble behind a linear embedding)
        self._gen = 0               # bumped by refresh()
        self._workspace = None
        self._fast = None           # what the last blocking one-query shadow(cuda=True) was prepared for (_shadow_prepared)
        self.last_profile = None
        self.last_path = None       # "hip" / "torch": which implementation served the last shadow() / predict()
        self.last_predict_reduction = None   # device_predict: "device" (psh_weighted_moments) / "host" (the class's own avg / std)

    @staticmethod
    def _load_with_scatspectra(dataset):
        # ref :84-88 -- Path / TimeSeriesDataset inputs go through the un-vendored scatspectra
        try:
            from scatspectra import TimeSeriesDataset  # type: ignore
        except Exception as e:  # noqa: BLE001
            # without the un-vendored package: a directory of the reference's own batch files
            # (scripts/batch_generations.py: batchNNNN.npy) is read directly
            if isinstance(dataset, Path) and dataset.is_dir():
                from . import ingest
                return ingest.load_batches(dataset)[0]
            raise ImportError("loading a dataset from a path needs the `scatspectra` package, or a directory of "
                              "batchNNNN.npy files (pass the array itself instead)") from e
        if isinstance(dataset, Path):
            dataset = TimeSeriesDataset(dpath=dataset, R=None).load()
        if isinstance(dataset, TimeSeriesDataset):
            dataset = dataset.load()
        return dataset

    def _dataset_tensor(self) -> torch.Tensor:
        """The ensemble as a float32 (R, C, T) torch tensor, read afresh on every call like the reference
        (ref :205) -- nothing converted is kept on the host.  A contiguous float32 numpy array is wrapped
        without a copy (the reference copies 0.5 GB per call at R = 32768); any other dtype / layout is
        converted per call."""
        ds = self.dataset
        if isinstance(ds, torch.Tensor):
            return _dim_array(ds)
        arr = _dim_array(np.asarray(ds))
        if arr.dtype == np.float32 and arr.flags.c_contiguous:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)   # a read-only array is wrapped as is: nothing here writes to it
                return torch.as_tensor(arr)
        return torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float32))

    def refresh(self) -> None:
        """Forget the copy of `dataset` resident in HBM: call after editing the array in place when the object
        was built with cache=True (with cache="auto" a writeable numpy array is re-read on every call anyway)."""
        self._resident = self._scan_rows = self._dirty_split = None
        self._gen += 1

    def _may_keep_resident(self) -> bool:
        if self.cache is True:
            return True
        if self.cache is False:
            return False
        ds = self.dataset
        if isinstance(ds, torch.Tensor):
            # keyed on the tensor's version counter, which every in-place TORCH op bumps.  (A CPU tensor made by
            # torch.from_numpy can still be edited through the numpy array without a bump: callers who do that pass
            # cache=False or call refresh() -- documented in README.)
            return True
        arr = ds if isinstance(ds, np.ndarray) else None
        if arr is None:
            return False                              # lists etc.: converted (and re-read) per call
        # read-only all the way down: a read-only VIEW of a writeable base (arr.view() with writeable=False,
        # np.broadcast_to) can still change through its base
        while isinstance(arr, np.ndarray):
            if arr.flags.writeable:
                return False
            arr = arr.base
        import mmap
        return arr is None or isinstance(arr, (bytes, mmap.mmap))   # owns its data / immutable bytes / a file mapped read-only

    # ------------------------------------------------------------------ native path
    def _native_kind(self, x: torch.Tensor, y: torch.Tensor, k: int) -> str | None:
        """Which HIP path implements this configuration, if any (types compared with
        `is`, so user subclasses keep their own behaviour on the generic path):
          "identity": Identity + RelativeMSE + PredictionContext -- windows read in place;
          "linear"  : any other PathEmbedding whose forward is the stock conv1d (Foveal,
                      PathEmbedding(kernel)) + RelativeMSE + PredictionContext, one-window
                      queries, kernel small enough for LDS -- psh_scan_topk_embedded;
          "padded"  : a stock embedding (Identity included) + RelativeMSE + ImputationContext((l, c, r)):
                      the context's zero taps over the gap are part of the scanning kernel -- the same
                      entry point with the padded (d, l+c+r) kernel and no trailing horizon.
        A CrossChannelContext(oc) (ensemble of 1 + oc channels, query on the first) pads the kernel with zero
        taps on the other channels: the scan is the "identity" / "linear" one over channel 0, the gathered paths
        keep every channel."""
        n_ch = 1
        if type(self.context) is CrossChannelContext and isinstance(self.context.out_context_channels, int):
            n_ch = 1 + self.context.out_context_channels
        if not (type(self.distance) is RelativeMSE and x.shape[1] == 1 and y.ndim == 3 and y.shape[1] == n_ch
                and x.dtype == torch.float32 and y.dtype == torch.float32 and k <= _native.PSH_MAX_K):
            return None
        emb = self.embedding
        stock = (isinstance(emb, PathEmbedding) and type(emb).forward is PathEmbedding.forward
                 and type(emb).adjust_to_context is PathEmbedding.adjust_to_context
                 and emb.kernel.ndim == 3 and emb.kernel.shape[1] == 1 and emb.kernel.dtype == torch.float32
                 and emb.kernel.shape[-1] == x.shape[-1])
        if type(self.context) is ImputationContext:
            p = self.context.portion
            if (stock and p is not None and len(p) == 3 and min(p) > 0 and p[0] + p[2] == x.shape[-1]
                    and _native.embedding_supported(emb.kernel.shape[0], sum(p))):
                return "padded"
            return None
        if type(self.context) not in (PredictionContext, CrossChannelContext):
            return None
        if type(self.context) is CrossChannelContext and n_ch < 2:
            return None
        if type(emb) is Identity:
            ok = x.shape[-1] == emb.kernel.shape[0] and x.shape[-1] <= _native.PSH_MAX_W
            return "identity" if ok else None
        if stock and _native.embedding_supported(emb.kernel.shape[0], emb.kernel.shape[-1]):
            return "linear"
        return None

    def _native_ok(self, x: torch.Tensor, y: torch.Tensor, k: int) -> bool:
        return self._native_kind(x, y, k) is not None

    @staticmethod
    def _hip_device() -> torch.device:
        if not torch.cuda.is_available():
            raise _native.NativeLibraryError(
                "cuda=True needs a HIP device (torch.cuda.is_available() is False); "
                "there is no CPU fallback on this path -- use cuda=False for the host path")
        return torch.device("cuda", torch.cuda.current_device())

    @staticmethod
    def _caller_stacklevel() -> int:
        """`stacklevel` that makes a warning point at the first frame OUTSIDE this module (shadow(), predict() and the device-side
        predict reach the warning through different depths)."""
        import sys
        level, f = 1, sys._getframe(1)
        while f is not None and f.f_globals.get("__name__") == __name__:
            f = f.f_back
            level += 1
        return level

    def _resident_dataset(self, y: torch.Tensor, device: torch.device) -> torch.Tensor:
        """The (R, C, T) ensemble in HBM, uploaded once and kept while `y` is the same
        host storage (the reference re-uploads every split on every call, ref :154-155)."""
        if y.is_cuda:
            return y.contiguous()
        if not self._may_keep_resident():
            # a dataset that may change between calls without a trace (a writeable numpy array -- what the tutorial and
            # TimeSeriesDataset.load() hand over) is uploaded on every call, as the reference does (ref :154-155).  Two things
            # keep that from being paid silently: the calls of ONE predict() share one upload (the array cannot change between
            # its context splits: ref :286-301 loops over them without returning to the caller), and the first upload warns,
            # with the remedy.
            scope = self._predict_scope
            if scope is not None and scope[0] is self.dataset and scope[1] is not None and scope[1].device == device:
                return scope[1]
            self._resident = self._scan_rows = self._dirty_split = None
            if not self._warned_reupload:
                self._warned_reupload = True
                mib = y.numel() * 4 / 2 ** 20
                warnings.warn(f"PathShadowing(cuda=True): the dataset is a writeable numpy array, so it is uploaded to the GPU again on "
                              f"every call ({mib:.0f} MiB each time, as the reference does) -- construct with cache=True (and call "
                              "refresh() after editing the array in place), or pass a read-only array or a torch tensor, to keep it "
                              "resident in HBM", RuntimeWarning, stacklevel=self._caller_stacklevel())
            up = y.contiguous().to(device, non_blocking=False)
            if scope is not None and scope[0] is self.dataset:
                self._predict_scope = (self.dataset, up)
            return up
        # keyed on the host storage AND the identity of what owns it: for a torch dataset the tensor object itself
        # (a weak reference: an address the allocator reuses for another tensor is not a match), for a numpy
        # dataset the array object
        owner = self.dataset
        key = (y.data_ptr(), tuple(y.shape), y._version if isinstance(owner, torch.Tensor) else 0, device, self._gen)
        hit = (self._resident is not None and self._resident[0] == key and self._resident[2]() is owner)
        if not hit:
            try:
                ref = weakref.ref(owner)
            except TypeError:                         # (a list: no weak references; cache=True only)
                ref = (lambda o=owner: o)
            self._resident = (key, y.contiguous().to(device, non_blocking=False), ref)
            self._scan_rows = None
        return self._resident[1]

    def _scan_rows_of(self, ds: torch.Tensor, smear: bool = True) -> torch.Tensor:
        """(R, T) rows the scan reads: the ensemble itself, or -- several channels, CrossChannelContext -- a
        contiguous copy of channel 0 kept beside it.  An ensemble that holds NaN / +-inf samples (looked for ONCE per
        resident copy: psh_count_nonfinite; `self._dirty` afterwards) is scanned by the Identity scan through rows in which
        those samples are written back over the horizon before them, so that a window is NaN exactly where the reference's
        zero-padded conv makes it NaN -- a non-finite sample anywhere in y[r, :, t : t+W+h] (ref path_embedding.py:48-51,
        :129-132; psh_prep.hip); paths are still gathered from the ensemble itself.  (The embedded scans look at the taps
        their kernel rows span, which no rewriting of the data turns into the conv's rule for every kernel: a dirty ensemble
        behind a linear embedding takes the generic torch formulation, _native_scan.)"""
        back = int(self.context.get_out_times())
        # keyed on the STORAGE (address, shape, version counter), not on the tensor object: _dim_array hands a fresh
        # ds[:, None, :] view of a 2-D CUDA dataset on every call, and a key that wanted the same object recounted the whole
        # ensemble -- a pass over it plus a host synchronisation -- on every shadow() call.  The cache keeps `ds` alive, so the
        # address cannot be handed to another tensor while the entry exists.
        # (`smear` False: a linear embedding's scan never reads the smeared copy -- its dirty rows go through _split_dirty_rows --
        #  so a dirty ensemble does not pay for a second R x T array it would not use)
        key = (ds.data_ptr(), tuple(ds.shape), tuple(ds.stride()), ds._version, back, str(ds.device), bool(smear))
        if self._scan_rows is None or self._scan_rows[0] != key:
            if self._scan_rows is None or self._scan_rows[0][:6] != key[:6]:
                self._dirty = bool(_native.count_nonfinite(ds))
                self._dirty_split = None
            if self._dirty and smear:
                rows = _native.smear_nonfinite(ds, back, 0)
            else:
                rows = ds[:, 0, :] if ds.shape[1] == 1 else ds[:, 0, :].contiguous()
            self._scan_rows = (key, rows, ds)
        return self._scan_rows[1]

    def _split_dirty_rows(self, ds: torch.Tensor, back: int):
        """A dirty ensemble behind a linear embedding, once per resident copy: (clean row indices, their rows (Rc, T) as a
        contiguous copy, dirty row indices, THEIR rows with every non-finite sample written back over the `back` samples before
        it).  The embedded scans' rejection tests assume finite data (prefix sums and matrix-core tiles spread a NaN over clean
        windows); the dense chains of the exhaustive path multiply all K taps, zeros included, and so meet a NaN exactly where
        the reference's zero-padded conv does (ref path_embedding.py:48-51, :129-132) once the horizon is smeared in."""
        if getattr(self, "_dirty_split", None) is None or self._dirty_split[0] != back:
            flags = _native.rows_nonfinite(ds)
            dirty_idx = torch.nonzero(flags).flatten()
            clean_idx = torch.nonzero(flags == 0).flatten()
            clean_rows = ds[clean_idx, 0, :].contiguous()
            dirty_rows = _native.smear_nonfinite(ds[dirty_idx].contiguous(), back, 0) if dirty_idx.numel() else None
            self._dirty_split = (back, clean_idx, clean_rows, dirty_idx, dirty_rows)
        return self._dirty_split[1:]

    def _native_scan(self, x: torch.Tensor, y: torch.Tensor, k: int, defer_status: bool = False):
        """(d, idx, resident dataset) on the device.  `defer_status` (Identity scans only): ONE raw call, its status
        tensor returned as a fourth item instead of being waited for -- the caller reads it together with the results
        (one synchronisation per call instead of two) and comes back without the flag when it is not zero."""
        dev = self._hip_device()
        _native.load()
        self._served_by = "hip"
        ds = self._resident_dataset(y, dev)
        rows = self._scan_rows_of(ds, smear=self._native_kind(x, y, k) not in ("linear", "padded"))
        h = self.context.get_out_times()
        if self._workspace is None or self._workspace.device != dev:
            self._workspace = _native.Workspace(dev)
        n_windows = ds.shape[0] * (ds.shape[-1] - x.shape[-1] - h + 1)
        if k > n_windows:
            # the reference fails inside torch.topk (ref :165) with the same exception type
            raise RuntimeError("selected index k out of range")
        kind = self._native_kind(x, y, k)
        if kind in ("linear", "padded"):
            # the (tiny) query embedding stays the module's own conv1d (ref :140); the scan
            # over the ensemble takes the unpadded kernel and the horizon as an integer --
            # or, for an ImputationContext, the kernel with the gap's zero taps and no horizon
            ker = self.embedding.kernel
            with single_thread():
                hx = self.embedding(x.to(ker.device))[:, 0, :].contiguous()
            hx = hx.to(dev)
            back = h
            if kind == "padded":
                h = 0
            # the scanning kernel on the device, kept while the module's kernel tensor is the same object at the same
            # version (an in-place edit bumps it): no upload per call, and the library may keep what it found in the matrix
            kkey = (id(ker), ker._version, kind, str(dev), tuple(self.context.portion) if kind == "padded" else None)
            if getattr(self, "_ker_dev", None) is None or self._ker_dev[0] != kkey:
                if kind == "padded":
                    ker2 = self.context.pad_context(ker)[:, 0, :].contiguous().to(dev)
                else:
                    ker2 = ker[:, 0, :].contiguous().to(dev)
                self._ker_dev = (kkey, ker2, ker)            # (ker: keeps the id alive)
            ker2 = self._ker_dev[1]
            # a kernel without Foveal's suffix structure (a filter bank, a user kernel): the rejection test on the
            # matrix cores (the library cannot look at the matrix without a synchronisation: the caller says which it is)
            fl = 0 if isinstance(self.embedding, Foveal) else _native.FLAG_EMBED_MX

            def embedded_topk(rows_t, kk, keep_plan):
                """(d, idx) of the kk best windows of `rows_t` per query: the sampled scan, the exhaustive one for queries
                whose status asks for it (candidate slices overflowed: massive ties / adversarial data)."""
                d, idx, status = _native.scan_topk_embedded(rows_t, ker2, hx, kk, h=h, workspace=self._workspace, flags=fl,
                                                            keep_plan=keep_plan)
                bad = torch.nonzero(status != _native.PSH_STATUS_OK).flatten()
                if bad.numel():
                    d2, idx2, _ = _native.scan_topk_embedded(rows_t, ker2, hx[bad].contiguous(), kk, h=h, workspace=self._workspace,
                                                             exhaustive=True, flags=fl)
                    d[bad] = d2
                    idx[bad] = idx2
                return d, idx

            one_window = rows.shape[-1] == ker2.shape[-1] + h
            if not self._dirty:
                if one_window:
                    # ONE window per row: the reference's embedded view (S, 1, d) is contiguous and its distance the
                    # 8-lane reduce over d (ref path_embedding.py:129-132, path_distance.py:65) -- embed every row once,
                    # then scan R pre-embedded points (rows one window long): psh_embed_rows + psh_scan_topk
                    points = _native.embed_rows(rows.contiguous(), ker2)
                    d, idx = _native.scan_topk_checked(points, hx, k, h=0, workspace=self._workspace)
                    return d, idx, ds
                # (keep_plan: ker2 is the cached device copy above -- same tensor, same version = same matrix)
                d, idx = embedded_topk(rows, k, True)
                return d, idx, ds
            # ---- NaN / +-inf in the ensemble behind a linear embedding.  The reference's conv1d makes a window NaN when ANY tap
            # of its zero-padded kernel meets one (0 * NaN); the native rejection tests assume finite data.  Rows without such a
            # sample (almost all) are scanned as ever; the few that hold one go through the exhaustive dense chains on rows
            # with the horizon smeared in (every tap multiplied: the conv's rule exactly); the two lists are merged by (d, r, t).
            clean_idx, clean_rows, dirty_idx, dirty_rows = self._split_dirty_rows(ds, back if kind == "linear" else 0)
            Tp = ds.shape[-1] - ker2.shape[-1] - h + 1
            B_ = hx.shape[0]
            parts_d, parts_i = [], []
            n_clean, n_dirty = int(clean_idx.numel()) * Tp, int(dirty_idx.numel()) * Tp
            if n_clean > 0:
                kc = min(k, n_clean)
                if one_window:
                    points = _native.embed_rows(clean_rows, ker2)
                    dc, ic = _native.scan_topk_checked(points, hx, kc, h=0, workspace=self._workspace)
                elif kc == k:
                    dc, ic = embedded_topk(clean_rows, kc, False)
                else:
                    dc, ic, _ = _native.scan_topk_embedded(clean_rows, ker2, hx, kc, h=h, workspace=self._workspace, exhaustive=True, flags=fl)
                ic = ic.clone()
                ic[..., 0] = clean_idx[ic[..., 0].long()].to(torch.int32)
                parts_d.append(dc)
                parts_i.append(ic)
            if n_dirty > 0 and (n_clean < k or not one_window):
                kd = min(k, n_dirty)
                if one_window:
                    # a dirty row IS its one window: NaN, ranked behind every clean one (ref :165)
                    dd = torch.full((B_, kd), float("nan"), dtype=torch.float32, device=dev)
                    idd = torch.zeros((B_, kd, 2), dtype=torch.int32, device=dev)
                    idd[..., 0] = dirty_idx[:kd].to(torch.int32)[None, :]
                else:
                    # (EMBED_DENSE: every one of the K taps is multiplied, zeros included -- the suffix-rows walk would skip the
                    #  taps in front of Foveal's longest row, where the conv still meets a NaN)
                    dd, idd, _ = _native.scan_topk_embedded(dirty_rows, ker2, hx, kd, h=h, workspace=self._workspace, exhaustive=True,
                                                            flags=_native.FLAG_EMBED_DENSE)
                    idd = idd.clone()
                    idd[..., 0] = dirty_idx[idd[..., 0].long()].to(torch.int32)
                parts_d.append(dd)
                parts_i.append(idd)
            if len(parts_d) == 1 and parts_d[0].shape[1] == k:
                return parts_d[0], parts_i[0], ds
            d, idx = _native.merge_topk(torch.cat(parts_d, dim=1).contiguous(), torch.cat(parts_i, dim=1).contiguous(), k)
            return d, idx, ds

        # ---- Identity windows
        # A batch's 8-bit rejection test (32 queries and more, W <= 25: psh_capi.hip) puts every query of a call on ONE
        # quantisation step (include/psh.h, PSH_FLAG_MQ_F16): queries that differ in amplitude by more than ~3x go to the library
        # as separate calls, one per amplitude class (a factor of 3 each) -- a call's time is proportional to its queries, so the
        # classes cost what the batch would, plus a quarter of a millisecond of fixed work per class -- as long as every class
        # keeps the 32 queries the 8-bit test wants.  Otherwise (small classes, small batches, longer windows: the f16 test) the
        # classes are a factor of 64 wide.  Decided here, where the queries are still host memory.
        classes, flags = None, 0
        i8 = x.shape[0] >= 32 and x.shape[-1] <= 25            # the call would meet the 8-bit test
        if x.shape[0] > 1 and x.device.type == "cpu":
            amp = x[:, 0, :].abs().amax(dim=1)
            top = float(amp[torch.isfinite(amp)].max()) if bool(torch.isfinite(amp).any()) else 0.0
            if top > 0.0 and not (top <= 3.0 * float(amp.min())):
                def by_factor(f):
                    cls = torch.floor(torch.log(torch.clamp(amp / top, min=1e-30)) / math.log(f) + 1e-6).to(torch.int64)
                    cls = torch.where(torch.isfinite(amp) & (amp > 0), cls, torch.full_like(cls, -1000))   # zero / non-finite queries: a class of their own
                    return [torch.nonzero(cls == c).flatten() for c in torch.unique(cls, sorted=True).tolist()[::-1]]
                fine = by_factor(3.0)
                if i8 and min(int(c.numel()) for c in fine) >= 32:
                    classes = fine                                # every class keeps the 8-bit test
                else:
                    # the f16 test copes with amplitudes a few dozen times apart on its one scale (beyond that a quiet query
                    # keeps more windows than its slices hold and falls to the exhaustive pass): classes a factor of 64 wide,
                    # each ONE call, on the f16 test where the 8-bit one would have met more than its factor of 3
                    classes = by_factor(64.0)
                    flags = _native.FLAG_MQ_F16 if i8 else 0
                if len(classes) == 1:
                    classes = None
        if classes is None:
            xq = x[:, 0, :].contiguous().to(dev)
            if defer_status:
                d, idx, status = _native.scan_topk(rows, xq, k, h=h, workspace=self._workspace, flags=flags)
                return d, idx, ds, status
            d, idx = _native.scan_topk_checked(rows, xq, k, h=h, workspace=self._workspace, flags=flags)
            return d, idx, ds
        B_ = x.shape[0]
        d = torch.empty((B_, k), dtype=torch.float32, device=dev)
        idx = torch.empty((B_, k, 2), dtype=torch.int32, device=dev)
        status = torch.zeros(B_, dtype=torch.int32, device=dev)
        for sel in classes:
            xq = x[sel, 0, :].contiguous().to(dev)
            sel_d = sel.to(dev)
            if defer_status:
                # (no synchronisation per class: the statuses travel with the results, the caller reads them once)
                dc, ic, sc = _native.scan_topk(rows, xq, k, h=h, workspace=self._workspace, flags=flags)
                status[sel_d] = sc
            else:
                dc, ic = _native.scan_topk_checked(rows, xq, k, h=h, workspace=self._workspace, flags=flags)
            d[sel_d] = dc
            idx[sel_d] = ic
        if defer_status:
            return d, idx, ds, status
        return d, idx, ds

    @staticmethod
    def _to_host(*tensors: torch.Tensor) -> tuple[np.ndarray, ...]:
        """Device results -> numpy through PINNED staging buffers, one synchronisation for
        all of them (the gathered paths are B*k*(T_x+h) floats -- 74 MB for the tutorial's
        call -- and a pageable .cpu() moves them at a third of the PCIe rate).  The arrays
        own their pinned blocks; torch's host allocator recycles them when they die."""
        host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
        for h_, t in zip(host, tensors):
            h_.copy_(t, non_blocking=True)
        torch.cuda.current_stream(tensors[0].device).synchronize()
        return tuple(h_.numpy() for h_ in host)

    # ------------------------------------------------------------------ generic path
    def _generic_scan(self, x: torch.Tensor, y: torch.Tensor, k: int, n_splits: int, cuda: bool):
        """The reference's formulation with stock torch ops (ref :129-177): embed the
        query, pad the kernel by the context, then per dataset split: embed, broadcast
        distance, top-k, decode, merge with the running best."""
        embedding, distance = self.embedding, self.distance
        dev = self._hip_device() if cuda else torch.device("cpu")
        x = x.to(dev)
        if cuda:
            # COPIES on the device: nn.Module.to() moves a module in place, and the caller's embedding must stay where it is
            # (its kernel's device decides where later calls embed their queries -- another device, another rounding)
            import copy
            embedding, distance = copy.deepcopy(embedding), copy.deepcopy(distance)
        embedding = embedding.to(dev)
        distance = distance.to(dev)
        n_query, n_paths = x.shape[0], y.shape[0]
        dim = embedding.kernel.shape[0] or x.shape[-1]
        hx = embedding(x)[:, 0, :]
        scanner = embedding.adjust_to_context(self.context)
        best_d = hx.new_full((n_query, k), float("inf"))
        best_i = torch.full((n_query, k, y.ndim - 1), -1, dtype=torch.int32, device=dev)
        hx = hx.view((n_query,) + (1,) * (y.ndim - 1) + (dim,))
        for rows in torch.arange(n_paths, dtype=torch.int32).split(n_paths // n_splits):
            hy = scanner(y[rows.long(), ...].to(dev))
            rows = rows.to(dev)
            dist = distance(hx, hy[None, ...]).view(n_query, -1)
            d_new, flat = torch.topk(dist, k=k, dim=-1, largest=False)
            axes = [rows] + [torch.arange(s, dtype=torch.int32, device=dev) for s in hy.shape[1:-1]]
            i_new = select_cartesian_product(flat.to(torch.int32), axes)
            pool_d = torch.cat([best_d, d_new], dim=1)
            pool_i = torch.cat([best_i, i_new], dim=1)
            best_d, pos = torch.topk(pool_d, k=k, dim=-1, largest=False)
            best_i = torch.gather(pool_i, 1, pos.unsqueeze(-1).expand(-1, -1, pool_i.shape[-1]))
        return best_d.cpu(), best_i.cpu()

    # ------------------------------------------------------------------ public API
    def batched_distance(self, x: torch.Tensor, y: torch.Tensor, k: int, n_splits: int, cuda: bool
                         ) -> tuple[torch.Tensor, torch.Tensor]:
        """k smallest distances d(h(x), h(y)) and where they are (ref :97-179).

        x (B, C, T_x) queries, y (S, C, T) dataset.  Returns CPU tensors:
        distances (B, k) float32 ascending and indices (B, k, 2) int32 = [path, time].
        `n_splits` only bounds memory on the generic path; the HIP path streams the
        dataset once and ignores it.
        """
        if cuda and self._native_ok(x, y, k):
            d, idx, _ = self._native_scan(x, y, k)
            return d.cpu(), idx.cpu()
        return self._generic_scan(x, y, k, n_splits, cuda)

    def shadow(self, x_context: ArrayType, k: int = 1, n_splits: int = 1, cuda: bool = False
               ) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Distances (B,k), shadowing paths with their out-context (B,k,C,T_x+h) and
        indices (B,k,2) of the k closest windows of the dataset (ref :181-218)."""
        ksize = self.embedding.kernel.shape[-1]
        if ksize != 0 and ksize != x_context.shape[-1]:
            raise Exception("The embedding kernel should be of the same size as the context.")
        if cuda and self._fast is not None:
            got = self._shadow_fast(self._fast, x_context, k)
            if got is not None:
                return got
        x = _torch(_dim_array(x_context))
        y = self._dataset_tensor()
        length = x.shape[-1] + self.context.get_out_times()

        if cuda and x.shape[0] == 1 and self._native_kind(x, y, k) == "identity":
            # one Identity query: a prepared slot (argument lists, pinned staging, result buffers built once) on the caller's
            # stream, the fused single launch -- 214 -> 162 us per call at the benchmark size, the kernels' 100 included
            got = self._shadow_prepared(x, y, k)
            if got is not None:
                return got
        if cuda and self._native_ok(x, y, k):
            out = self._native_scan(x, y, k, defer_status=True)
            self.last_path = self._served_by
            if len(out) == 4:
                # the status travels with the results: gather and copies are enqueued behind the scan unconditionally
                # (a status other than OK -- the fused launch gave up, candidate slices overflowed -- is rare and then
                # costs the wasted gather), one synchronisation for everything
                d, idx, ds, status = out
                paths = _native.gather_paths(ds, idx, length)
                hd, hp, hi, hs = self._to_host(d, paths, idx, status)
                if not hs.any():
                    return hd, hp, hi
                if bool((hs == _native.PSH_STATUS_RETRY).any()):
                    self._workspace.arm()
                out = self._native_scan(x, y, k)
            d, idx, ds = out
            paths = _native.gather_paths(ds, idx, length)           # (B, k, C, len) on device
            return self._to_host(d, paths, idx)

        d, idx = self._generic_scan(x, y, k, n_splits, cuda)
        self.last_path = "torch"
        # gather dataset[r, :, t : t+len] for every (r, t)  (ref :211-216)
        offs = torch.arange(length, dtype=torch.int64)
        r = idx[..., 0].long()
        t = idx[..., 1].long()[..., None] + offs                     # (B, k, len)
        paths = y[r[..., None], :, t]                                # (B, k, len, C)
        return _numpy(d), _numpy(paths.permute(0, 1, 3, 2).contiguous()), _numpy(idx)

    def _shadow_prepared(self, x: torch.Tensor, y: torch.Tensor, k: int):
        """shadow() for ONE Identity query as one blocking library call (_native.BlockingShadow -> psh_shadow_blocking: the fused
        launch gathers the paths itself and writes everything into a pinned block the results are handed out of); None when the
        status says the general path has to serve the call.  Leaves `self._fast`: what the next call checks -- a dozen identity
        comparisons -- to come straight back here without the general argument handling (ref :181-218 re-reads the dataset
        on every call; here the question is only whether anything a call depends on has been replaced)."""
        if not (y.is_cuda or self._may_keep_resident()):
            return None                                               # (an ensemble re-uploaded per call: no slot to keep)
        dev = self._hip_device()
        _native.load()
        ds = self._resident_dataset(y, dev)
        rows = self._scan_rows_of(ds)
        h = self.context.get_out_times()
        W = x.shape[-1]
        if k > ds.shape[0] * (ds.shape[-1] - W - h + 1):
            raise RuntimeError("selected index k out of range")       # (the reference fails inside torch.topk, ref :165)
        if self._workspace is None or self._workspace.device != dev:
            self._workspace = _native.Workspace(dev)
        # (the slot is bound to the workspace BUFFER: another call of this object may have grown it since)
        wsb = self._workspace.get(_native.workspace_bytes(rows.shape[0], rows.shape[1], 1, W, h, k))
        key = (ds.data_ptr(), tuple(ds.shape), ds._version, W, k, h, str(dev), wsb.data_ptr())
        st = getattr(self, "_sync_slot", None)
        if st is None or st[0] != key:
            st = self._sync_slot = (key, _native.BlockingShadow(rows, ds, W, k, h, self._workspace, 0))
        owner = self.dataset
        self._fast = {"slot": st[1], "owner": owner, "version": owner._version if isinstance(owner, torch.Tensor) else None,
                      "numpy_ro": isinstance(owner, np.ndarray) and self.cache is not True,
                      "emb": self.embedding, "kernel": self.embedding.kernel, "dist": self.distance, "ctx": self.context, "h": h,
                      "k": k, "W": W, "dev": dev.index, "gen": self._gen, "wsbuf": wsb, "cache": self.cache, "hint": self.hint,
                      "ds": ds, "rows": rows}
        xq = x[0, 0, :].detach()
        return self._shadow_blocking_call(self._fast, xq.cpu().numpy() if xq.is_cuda else xq.numpy())

    def _shadow_fast(self, fs: dict, x_context, k: int):
        """The blocking call again when NOTHING it depends on has changed since `_shadow_prepared` built `fs` (same dataset
        object and version, same plugin objects, horizon, k, device, workspace buffer, no refresh()); else None."""
        owner = self.dataset
        if not (k == fs["k"] and owner is fs["owner"] and self.embedding is fs["emb"] and self.embedding.kernel is fs["kernel"]
                and self.distance is fs["dist"] and self.context is fs["ctx"] and self._gen == fs["gen"] and self.cache == fs["cache"]
                and self.hint == fs["hint"] and self._workspace is not None and self._workspace.buf is fs["wsbuf"]
                and self.context.get_out_times() == fs["h"] and torch.cuda.current_device() == fs["dev"]):
            return None
        if fs["version"] is not None:
            if owner._version != fs["version"]:
                return None
        elif fs["numpy_ro"] and owner.flags.writeable:
            return None
        if isinstance(x_context, np.ndarray):
            xq = x_context
        elif isinstance(x_context, torch.Tensor) and not x_context.is_cuda and not x_context.requires_grad:
            xq = x_context.numpy()
        else:
            return None
        W = fs["W"]
        if xq.size != W or xq.shape[-1] != W or xq.ndim > 3:
            return None                                               # (several queries, another length: the general path says what is wrong)
        return self._shadow_blocking_call(fs, xq.reshape(W))

    def _shadow_blocking_call(self, fs: dict, xq: np.ndarray):
        slot, k, W = fs["slot"], fs["k"], fs["W"]
        stream = torch._C._cuda_getCurrentRawStream(fs["dev"])
        xn2 = None
        if self.hint == "auto":
            x32 = xq if xq.dtype == np.float32 else xq.astype(np.float32)      # (a hint: fp32 is plenty, and one numpy call)
            xn2 = float(np.dot(x32, x32))
        hint = self._next_hint(xn2, k, W)
        status, res = slot.call(stream, xq, hint)
        self.last_path = "hip"
        self.last_hint = None
        if hint is not None and status != _native.PSH_STATUS_OK:
            # the hint fell short of k windows (or admitted more than the lists hold): the same call without it; no hints for
            # a while (twice as long after every miss in a row)
            self.last_hint = "short"
            hstate = self._hint_state
            hstate["fails"] = min(hstate["fails"] + 1, 5)
            hstate["skip"] = 4 << hstate["fails"]
            # (the fused launch says RETRY for "fewer than k found" with its header still armed; one a time-out disarmed says
            #  RETRY again below and takes the general path)
            status, res = slot.call(stream, xq, None)
        elif hint is not None:
            self.last_hint = "ok"
            self._hint_state["fails"] = 0
        if status != _native.PSH_STATUS_OK:
            if status == _native.PSH_STATUS_RETRY:
                self._workspace.arm()
            return None
        if self.hint == "auto":
            h = fs["h"]
            hstate = self._hint_state if (self._hint_state and self._hint_state["key"] == (k, W, h)) else {"key": (k, W, h), "skip": 0, "fails": 0}
            hstate["dk"] = float(res[0][0, k - 1]) if xn2 > 0 else None
            self._hint_state = hstate
        return res

    HINT_MARGIN = 1.15          # on acc: the k-th distance may come out 7 % above the previous call's before the hint falls short

    def _next_hint(self, xn2: float | None, k: int, W: int):
        """The admission level this call hands to the library (hint="auto"), or None: (d_k of the previous call x ||x||)^2 x
        HINT_MARGIN -- the RELATIVE k-th distance moves far less from one query date to the next than acc itself."""
        hstate = self._hint_state
        if self.hint != "auto" or hstate is None or hstate.get("dk") is None or hstate["key"] != (k, W, self.context.get_out_times()):
            return None
        if hstate["skip"] > 0:
            hstate["skip"] -= 1
            return None
        # (windows below a level: ~ level^(W / 2).  The margin that triples the admitted count at any window length, capped at
        #  HINT_MARGIN: 1.15 up to W = 15, 1.017 at W = 126 -- a long window's hint holds only if d_k moves by less than 1 %)
        level = hstate["dk"] ** 2 * xn2 * min(self.HINT_MARGIN, 3.0 ** (2.0 / W))
        return level if (level > 0.0 and math.isfinite(level)) else None

    def shadow_async(self, x_context: ArrayType, k: int = 1, streams: int = 3) -> "PendingShadow":
        """shadow(cuda=True) without waiting for the result: the call only ENQUEUES the scan and the path gather and returns a
        handle whose `.result()` gives shadow()'s triple.  For callers with independent queries in flight (a server; a loop
        over query dates whose results are consumed later): consecutive calls rotate over `streams` private HIP streams, and
        a single query (Identity + RelativeMSE + PredictionContext) runs there as the overlap-friendly launches
        (PSH_FLAG_OVERLAP, psh_stream.hip) -- the sample and ranking launches of one call run beside the scan of another, so
        a stream of calls costs ~85 us each at the benchmark size where blocking calls cost ~160 (launch + results).  The
        reference has no counterpart (its shadow() is blocking, ref :181-218); any configuration this path does not cover is
        served by shadow() itself at call time and handed back through the same handle."""
        ksize = self.embedding.kernel.shape[-1]
        if ksize != 0 and ksize != x_context.shape[-1]:
            raise Exception("The embedding kernel should be of the same size as the context.")
        x = _torch(_dim_array(x_context))
        y = self._dataset_tensor()
        if not (self._native_kind(x, y, k) == "identity" and x.shape[0] == 1):
            return PendingShadow(self, None, self.shadow(x_context, k, cuda=True), None)
        dev = self._hip_device()
        _native.load()
        ds = self._resident_dataset(y, dev)
        rows = self._scan_rows_of(ds)
        h = self.context.get_out_times()
        W = x.shape[-1]
        if k > ds.shape[0] * (ds.shape[-1] - W - h + 1):
            raise RuntimeError("selected index k out of range")       # (the reference fails inside torch.topk, ref :165)
        # prepared slots (argument lists, pinned staging, result buffers): rebuilt when anything they were built for changes
        key = (ds.data_ptr(), tuple(ds.shape), ds._version, W, k, h, int(streams), str(dev))
        st = getattr(self, "_async", None)
        if st is None or st["key"] != key:
            n = int(streams)
            cur = torch.cuda.current_stream(dev)
            st = self._async = {"key": key, "i": 0, "streams": [torch.cuda.Stream(dev) for _ in range(n)],
                                "ws": [_native.Workspace(dev) for _ in range(n)], "free": [], "n": n}
            for s_ in st["streams"]:
                s_.wait_stream(cur)                                   # (the resident ensemble may just have been uploaded there)
            st["slots"] = [[] for _ in range(n)]
        i = st["i"] % st["n"]
        st["i"] += 1
        # a free slot of this stream, or a new one (a slot is busy until its handle's result() has been taken)
        pool = st["slots"][i]
        slot = next((sl for sl in pool if not sl.busy), None)
        if slot is None:
            with torch.cuda.stream(st["streams"][i]):
                slot = _native.PreparedShadow(rows, ds, W, k, h, st["ws"][i], _native.FLAG_OVERLAP)
            slot.busy = False
            pool.append(slot)
        slot.busy = True
        slot.launch(st["streams"][i], x[:, 0, :])
        self.last_path = "hip"
        return PendingShadow(self, slot.event, slot, (x_context, k))

    @staticmethod
    def init_averaging_proba(proba_name: str, distances: np.ndarray, eta: float | None) -> DiscreteProba:
        """"uniform" or "softmax" averaging over the shadowing paths (ref :220-232)."""
        if proba_name == "uniform":
            return Uniform()
        if proba_name == "softmax":
            return Softmax(distances, eta)
        raise ValueError("Unrecognized averaging proba")

    def predict_from_paths(self, distances: np.ndarray, paths: np.ndarray, to_predict: Callable,
                           proba_name: str, eta: float | None) -> tuple[np.ndarray, np.ndarray]:
        """Weighted mean / std over the k paths of `to_predict(out-context)` (ref :234-254)."""
        future = self.context.select_out_context(paths)
        proba = self.init_averaging_proba(proba_name, distances[:, :, None], eta)
        values = to_predict(future)
        return proba.avg(values, axis=1), proba.std(values, axis=1)

    def _predict_on_device(self, x: torch.Tensor, y: torch.Tensor, k: int, to_predict: Callable,
                           proba_name: str, eta: float | None):
        """shadow() + predict_from_paths() with the PATHS kept on the GPU: the k paths of every query (74 MB for
        the tutorial's call) stay in HBM and `to_predict` is evaluated there on the device tensor of their
        out-context; what crosses PCIe are the (B, k) distances -- the installed DiscreteProba turns them into its weights
        on the host (ref :245-250: scatspectra's classes when that package is importable, the stand-ins of averaging.py
        otherwise; no formula of theirs is restated) -- and the (B, ...) moments: `avg` / `std` over the k paths
        (ref :251-252) are reduced on the device by psh_weighted_moments when the class's own avg / std are the weighted
        moments of weights it exposes (moment_weights checks that on a probe; `last_predict_reduction` says which way a
        call went); any other class receives the (B, k, ...) statistic on the host and reduces it itself.  Only called
        when the caller opted in (see predict())."""
        length = x.shape[-1] + self.context.get_out_times()

        def evaluate(d, idx, ds):
            paths = _native.gather_paths(ds, idx, length)
            values = to_predict(self.context.select_out_context(paths))
            if not (isinstance(values, torch.Tensor) and values.dim() >= 2 and tuple(values.shape[:2]) == tuple(d.shape)):
                raise TypeError("device_predict: to_predict must map the (B, k, C, h) torch tensor of out-context paths to a "
                                f"torch tensor (B, k, ...); got {type(values).__name__}"
                                f"{tuple(getattr(values, 'shape', ()))}")
            return values

        out = self._native_scan(x, y, k, defer_status=True)
        self.last_path = self._served_by
        d_host = values = None
        if len(out) == 4:                                   # Identity scan: the status is read with the results (see shadow())
            d, idx, ds, status = out
            values = evaluate(d, idx, ds)
            d_host, hs = self._to_host(d, status)
            if hs.any():
                if bool((hs == _native.PSH_STATUS_RETRY).any()):
                    self._workspace.arm()
                d_host = None
                out = self._native_scan(x, y, k)
        if d_host is None:
            d, idx, ds = out
            values = evaluate(d, idx, ds)
            (d_host,) = self._to_host(d)
        # what has crossed PCIe so far: the (B, k) distances.  The installed class turns them into weights on the host (its
        # formula is its own); if its avg / std ARE the weighted moments of those weights -- checked on a probe, the class
        # stays authoritative -- the (B, k, ...) statistic is reduced where it is (psh_weighted_moments) and only the
        # (B, ...) moments come back.  Any other class gets the statistic on the host, as before.
        proba = self.init_averaging_proba(proba_name, d_host[:, :, None], eta)
        self.last_predict_reduction = "host"
        if values.is_cuda and values.dtype == torch.float32:
            w = moment_weights(proba, values.shape[0], values.shape[1])
            if w is not None:
                wt = None if w is True else torch.from_numpy(w).to(values.device)
                mean, std = _native.weighted_moments(values.contiguous(), wt)
                self.last_predict_reduction = "device"
                return self._to_host(mean, std)
        v_host = self._to_host(values.contiguous())[0] if values.is_cuda else values.numpy()
        return proba.avg(v_host, axis=1), proba.std(v_host, axis=1)

    def predict(self, x_context: ArrayType, k: int, to_predict: Callable, eta: float | None = None,
                proba_name: str = "softmax", n_dataset_splits: int = 1, n_context_splits: int = 1,
                cuda: bool = False, device_predict: bool | None = None) -> tuple[np.ndarray, np.ndarray]:
        """shadow() + predict_from_paths() over `n_context_splits` batches of queries (ref :256-301).

        `to_predict` receives what the reference hands it: the numpy array of out-context paths.  OPT-IN, with
        cuda=True on a natively scanned configuration: `device_predict=True` -- or a callable that declares
        `to_predict.accepts_torch = True`, as `shadowing.realized_variance` does -- evaluates `to_predict` on the
        device tensor instead, so the paths never leave HBM (_predict_on_device).  A numpy-style callable is NOT
        silently given tensors: `x.std(-1)` is Bessel-corrected in torch and not in numpy."""
        x = _torch(_dim_array(x_context))
        n = x.shape[0]
        y = None
        if device_predict is None:
            device_predict = bool(getattr(to_predict, "accepts_torch", False))
        means, stds = [], []
        # (cuda=True: the context splits of this ONE call share one upload of a dataset that is otherwise re-read per call)
        self._predict_scope = (self.dataset, None) if cuda else None
        try:
            for rows in tqdm(torch.arange(n).split(n // n_context_splits)):
                if cuda and device_predict:
                    y = self._dataset_tensor() if y is None else y
                    if self._native_ok(x[rows, ...], y, k):
                        got = self._predict_on_device(x[rows, ...], y, k, to_predict, proba_name, eta)
                        means.append(got[0])
                        stds.append(got[1])
                        continue
                d, paths, _ = self.shadow(x[rows, ...], k, n_dataset_splits, cuda)
                m, s = self.predict_from_paths(d, paths, to_predict, proba_name, eta)
                means.append(m)
                stds.append(s)
        finally:
            if self._predict_scope is not None and self._predict_scope[1] is not None:
                self._scan_rows = self._dirty_split = None          # (they keep the per-call upload alive)
            self._predict_scope = None
        return np.concatenate(means), np.concatenate(stds)


_MOMENT_CLASSES: dict = {}      # averaging class -> its avg / std ARE the weighted moments of its `weights` (probed once)


def moment_weights(proba, B: int, k: int):
    """The (B, k) float64 weights of an averaging object IF its `avg` / `std` over axis 1 are the weighted moments
    sum_j w_j x_j and sqrt(sum_j w_j (x_j - mean)^2) of those weights: True for uniform weights 1/k, an array otherwise,
    None when the object does not expose weights or when ITS OWN avg / std disagree with the moments on a probe (then the
    caller lets the object reduce on the host: the class is the authority, nothing of it is restated)."""
    if not hasattr(proba, "weights"):
        return None
    w = proba.weights
    if w is not None:
        try:
            w = np.asarray(w, dtype=np.float64)
        except Exception:  # noqa: BLE001
            return None
        while w.ndim > 2 and w.shape[-1] == 1:
            w = w[..., 0]
        if w.shape != (B, k):
            return None
        w = np.ascontiguousarray(w)
    verdict = _MOMENT_CLASSES.get(type(proba))               # the probe below runs once per averaging CLASS
    if verdict is not None:
        return (True if w is None else w) if verdict else None
    wf = np.full((B, k), 1.0 / k) if w is None else w
    j = np.arange(B * k * 2, dtype=np.float64).reshape(B, k, 2)
    probe = np.cos(0.37 * j) + 0.01 * j / (B * k)
    try:
        a, sd = np.asarray(proba.avg(probe, axis=1)), np.asarray(proba.std(probe, axis=1))
    except Exception:  # noqa: BLE001
        return None
    m = (wf[:, :, None] * probe).sum(axis=1)
    v = np.sqrt((wf[:, :, None] * (probe - m[:, None, :]) ** 2).sum(axis=1))
    ok = (a.shape == m.shape and sd.shape == v.shape
          and bool(np.allclose(a, m, rtol=1e-10, atol=1e-13) and np.allclose(sd, v, rtol=1e-10, atol=1e-13)))
    _MOMENT_CLASSES[type(proba)] = ok
    return (True if w is None else w) if ok else None


class PendingShadow:
    """Handle of PathShadowing.shadow_async(): `.result()` -> (distances (B,k), paths (B,k,C,W+h), indices (B,k,2)) as numpy
    arrays, exactly what shadow() returns; `.done()` -> whether the device has finished (no waiting)."""

    def __init__(self, owner: PathShadowing, event, payload, call):
        self._owner, self._event, self._payload, self._call = owner, event, payload, call

    def done(self) -> bool:
        return self._event is None or self._event.query()

    def __del__(self):
        # a handle dropped without result(): its slot goes back to the pool once the device is done with it
        try:
            if self._call is not None and self._event is not None:
                self._event.synchronize()
                self._payload.busy = False
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def result(self):
        if self._event is None:
            return self._payload                                      # served by shadow() at call time
        if self._call is not None:
            self._event.synchronize()
            slot = self._payload
            hd, hp, hi, hs = slot.take()                              # (small results: copies; large ones: the pinned buffer itself)
            slot.busy = False
            if hs.any():                                              # the status protocol: rare; the blocking path copes
                x_context, k = self._call
                hd, hp, hi = self._owner.shadow(x_context, k, cuda=True)
            self._payload, self._call = (hd, hp, hi), None
        return self._payload
