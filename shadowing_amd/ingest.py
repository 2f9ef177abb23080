"""Dataset ingestion: the `.npy` batch files the reference's scripts/batch_generations.py writes
(`batchNNNN.npy`, 256 generated paths per file, concatenated along the path axis) -> one trajectory
ensemble, resident in HBM (new work: the reference loads through the un-vendored
scatspectra package and re-uploads every split on every call).

  * the files are memory-mapped: only the rows a rank owns are ever read;
  * rows go to the device through two pinned staging buffers on a side stream (the copy of chunk i+1
    overlaps the read of chunk i+2 from the page cache);
  * `shard=(rank, world)` loads this rank's contiguous block of rows only (shadowing_amd.distributed.
    shard_rows) and reports its global row offset -- what ShardedPathShadowing wants.
"""
from __future__ import annotations

import re
from pathlib import Path

import numpy as np
import torch

from .distributed import shard_rows

_BATCH = re.compile(r"^batch(\d+)\.npy$")


def list_batches(dirpath) -> list[Path]:
    """batchNNNN.npy files of a directory, in batch order."""
    d = Path(dirpath)
    found = sorted((int(m.group(1)), p) for p in d.iterdir() if (m := _BATCH.match(p.name)))
    if not found:
        raise FileNotFoundError(f"no batchNNNN.npy files in {d}")
    return [p for _, p in found]


def _open(path: Path) -> np.ndarray:
    a = np.load(path, mmap_mode="r")
    if a.ndim == 2:                      # (n, T): single channel
        a = a[:, None, :]
    if a.ndim != 3:
        raise ValueError(f"{path}: expected (paths, channels, time) or (paths, time), got {a.shape}")
    return a


def describe(dirpath) -> dict:
    """Shapes without reading the data: {'files', 'rows' (per file), 'R', 'C', 'T'}."""
    files = list_batches(dirpath)
    maps = [_open(p) for p in files]
    C, T = maps[0].shape[1:]
    for p, a in zip(files, maps):
        if a.shape[1:] != (C, T):
            raise ValueError(f"{p}: shape {a.shape} does not match (*, {C}, {T})")
    rows = [int(a.shape[0]) for a in maps]
    return {"files": files, "rows": rows, "R": int(sum(rows)), "C": int(C), "T": int(T)}


def load_batches(dirpath, device: torch.device | str | None = None, shard: tuple[int, int] | None = None,
                 chunk_rows: int = 4096):
    """Returns (ensemble, row_offset, R_total): `ensemble` is (R_local, C, T) float32 -- a CPU tensor when
    `device` is None, else resident on that device -- holding global rows [row_offset, row_offset + R_local)."""
    info = describe(dirpath)
    R, C, T = info["R"], info["C"], info["T"]
    lo, hi = (0, R) if shard is None else shard_rows(R, shard[1], shard[0])
    n = hi - lo
    maps = [_open(p) for p in info["files"]]
    starts = np.concatenate([[0], np.cumsum(info["rows"])])

    def rows(a: int, b: int) -> np.ndarray:                 # global rows [a, b) as float32, C-contiguous
        parts = []
        for f, m in enumerate(maps):
            s, e = max(a, starts[f]), min(b, starts[f + 1])
            if s < e:
                parts.append(np.array(m[s - starts[f]:e - starts[f]], dtype=np.float32))   # a copy: the map is read-only
        return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0)

    if device is None:
        out = torch.empty((n, C, T), dtype=torch.float32)
        for a in range(lo, hi, chunk_rows):
            b = min(hi, a + chunk_rows)
            out[a - lo:b - lo] = torch.from_numpy(np.ascontiguousarray(rows(a, b)))
        return out, lo, R
    dev = torch.device(device)
    out = torch.empty((n, C, T), dtype=torch.float32, device=dev)
    stage = [torch.empty((min(chunk_rows, max(n, 1)), C, T), dtype=torch.float32, pin_memory=True) for _ in range(2)]
    done = [torch.cuda.Event(), torch.cuda.Event()]
    side = torch.cuda.Stream(device=dev)
    # `out` may be a block the caching allocator recycled from work still queued on the current stream: the copies
    # on the side stream start after it
    side.wait_stream(torch.cuda.current_stream(dev))
    for i, a in enumerate(range(lo, hi, chunk_rows)):
        b = min(hi, a + chunk_rows)
        buf = stage[i & 1]
        if i >= 2:
            done[i & 1].synchronize()                        # the copy that last used this buffer
        buf[:b - a].copy_(torch.from_numpy(np.ascontiguousarray(rows(a, b))))
        with torch.cuda.stream(side):
            out[a - lo:b - lo].copy_(buf[:b - a], non_blocking=True)
            done[i & 1].record(side)
    torch.cuda.current_stream(dev).wait_stream(side)
    side.synchronize()                                       # the staging buffers die with this frame
    return out, lo, R
