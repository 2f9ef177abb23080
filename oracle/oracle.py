"""ctypes front-end of oracle/libpsh_oracle.so (test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "libpsh_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    """Compile psh_oracle.c with the committed Makefile (gcc, a second or two)."""
    src = _HERE / "psh_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B"], check=True, capture_output=True)
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        L = C.CDLL(str(_SO))
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.psh_oracle_qnorm.restype = C.c_float
        L.psh_oracle_qnorm.argtypes = [f32p, C.c_int]
        L.psh_oracle_scan_topk.restype = C.c_int
        L.psh_oracle_scan_topk.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int64, f32p, f32p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, f32p, i32p, C.c_int]
        L.psh_oracle_all_distances.restype = C.c_int
        L.psh_oracle_all_distances.argtypes = [f32p, C.c_int64, C.c_int64, f32p, C.c_float,
                                               C.c_int, C.c_int, f32p]
        L.psh_oracle_gather_paths.restype = C.c_int
        L.psh_oracle_gather_paths.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int64, i32p,
                                              C.c_int64, C.c_int, f32p]
        L.psh_oracle_scan_topk_embedded.restype = C.c_int
        L.psh_oracle_scan_topk_embedded.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int64, f32p, C.c_int, C.c_int,
                                                    f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, i32p, C.c_int]
        L.psh_oracle_all_distances_embedded.restype = C.c_int
        L.psh_oracle_all_distances_embedded.argtypes = [f32p, C.c_int64, C.c_int64, f32p, C.c_int, C.c_int,
                                                        f32p, C.c_float, C.c_int, f32p]
        L.psh_oracle_all_acc.restype = C.c_int
        L.psh_oracle_all_acc.argtypes = [f32p, C.c_int64, C.c_int64, f32p, C.c_int, C.c_int, f32p]
        L.psh_oracle_all_acc_embedded.restype = C.c_int
        L.psh_oracle_all_acc_embedded.argtypes = [f32p, C.c_int64, C.c_int64, f32p, C.c_int, C.c_int, f32p, C.c_int, f32p]
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _rows(dataset) -> np.ndarray:
    ds = _f32(dataset)
    if ds.ndim == 3:
        if ds.shape[1] != 1:
            raise ValueError("oracle covers the single-channel path only")
        ds = ds[:, 0, :]
    elif ds.ndim == 1:
        ds = ds[None, :]
    return np.ascontiguousarray(ds)


def qnorm(queries) -> np.ndarray:
    """||x|| per query in the reference's reduction order (path_distance.py:65)."""
    q = np.atleast_2d(_f32(queries))
    return np.array([lib().psh_oracle_qnorm(_p(np.ascontiguousarray(r), C.c_float), q.shape[1])
                     for r in q], dtype=np.float32)


def scan_topk(dataset, queries, k: int, h: int = 0, r_offset: int = 0, qn=None,
              nthreads: int = 0):
    """(d (B,k) f32, idx (B,k,2) i32), canonical order (d, r, t) ascending."""
    ds = _rows(dataset)
    q = np.atleast_2d(_f32(queries))
    B, W = q.shape
    R, T = ds.shape
    d = np.empty((B, k), np.float32)
    idx = np.empty((B, k, 2), np.int32)
    qn_arr = None if qn is None else _f32(qn).reshape(B)
    rc = lib().psh_oracle_scan_topk(_p(ds, C.c_float), R, T, r_offset, _p(q, C.c_float),
                                    None if qn_arr is None else _p(qn_arr, C.c_float),
                                    B, W, h, k, _p(d, C.c_float), _p(idx, C.c_int32), nthreads)
    if rc != 0:
        raise RuntimeError(f"psh_oracle_scan_topk failed: {rc}")
    return d, idx


def all_distances(dataset, query, h: int = 0, qn=None) -> np.ndarray:
    ds = _rows(dataset)
    x = _f32(query).reshape(-1)
    R, T = ds.shape
    W = x.shape[0]
    Tp = T - W - h + 1
    out = np.empty((R, Tp), np.float32)
    xn = float(qnorm(x)[0]) if qn is None else float(qn)
    rc = lib().psh_oracle_all_distances(_p(ds, C.c_float), R, T, _p(x, C.c_float), xn, W, h,
                                        _p(out, C.c_float))
    if rc != 0:
        raise RuntimeError("psh_oracle_all_distances failed")
    return out


def all_acc(dataset, query, h: int = 0) -> np.ndarray:
    """(R, T') float32: every window's numerator acc = the sequential fp32 chain sum_j (x_j - y_{t+j})^2, before the
    square root and the division by ||x|| -- what the scans compare with their admission level."""
    ds = _rows(dataset)
    x = _f32(query).reshape(-1)
    R, T = ds.shape
    W = x.shape[0]
    out = np.empty((R, T - W - h + 1), np.float32)
    if lib().psh_oracle_all_acc(_p(ds, C.c_float), R, T, _p(x, C.c_float), W, h, _p(out, C.c_float)) != 0:
        raise RuntimeError("psh_oracle_all_acc failed")
    return out


def all_acc_embedded(dataset, kernel, hx, h: int = 0) -> np.ndarray:
    """(R, T') float32: every window's embedded numerator sum_i (hx_i - hy_i)^2 in the oracle's order."""
    ds = _rows(dataset)
    ker = _f32(kernel)
    if ker.ndim == 3:
        ker = ker[:, 0, :]
    ker = np.ascontiguousarray(ker)
    d_, K = ker.shape
    x = _f32(hx).reshape(-1)
    R, T = ds.shape
    out = np.empty((R, T - K - h + 1), np.float32)
    if lib().psh_oracle_all_acc_embedded(_p(ds, C.c_float), R, T, _p(ker, C.c_float), d_, K, _p(x, C.c_float), h,
                                         _p(out, C.c_float)) != 0:
        raise RuntimeError("psh_oracle_all_acc_embedded failed")
    return out


def scan_topk_embedded(dataset, kernel, hx, k: int, h: int = 0, r_offset: int = 0, hxnorm=None,
                       nthreads: int = 0):
    """Embedded scan: kernel (d, K) or (d, 1, K) unpadded, hx (B, d) embedded queries.
    (d (B,k) f32, idx (B,k,2) i32) in canonical order."""
    ds = _rows(dataset)
    ker = _f32(kernel)
    if ker.ndim == 3:
        ker = ker[:, 0, :]
    ker = np.ascontiguousarray(ker)
    d_, K = ker.shape
    q = np.atleast_2d(_f32(hx))
    B = q.shape[0]
    assert q.shape[1] == d_
    R, T = ds.shape
    d = np.empty((B, k), np.float32)
    idx = np.empty((B, k, 2), np.int32)
    qn_arr = None if hxnorm is None else _f32(hxnorm).reshape(B)
    rc = lib().psh_oracle_scan_topk_embedded(_p(ds, C.c_float), R, T, r_offset, _p(ker, C.c_float), d_, K,
                                             _p(q, C.c_float),
                                             None if qn_arr is None else _p(qn_arr, C.c_float),
                                             B, h, k, _p(d, C.c_float), _p(idx, C.c_int32), nthreads)
    if rc != 0:
        raise RuntimeError(f"psh_oracle_scan_topk_embedded failed: {rc}")
    return d, idx


def all_distances_embedded(dataset, kernel, hx, h: int = 0, hxnorm=None) -> np.ndarray:
    ds = _rows(dataset)
    ker = _f32(kernel)
    if ker.ndim == 3:
        ker = ker[:, 0, :]
    ker = np.ascontiguousarray(ker)
    d_, K = ker.shape
    x = _f32(hx).reshape(-1)
    R, T = ds.shape
    Tp = T - K - h + 1
    out = np.empty((R, Tp), np.float32)
    xn = float(qnorm(x)[0]) if hxnorm is None else float(hxnorm)
    rc = lib().psh_oracle_all_distances_embedded(_p(ds, C.c_float), R, T, _p(ker, C.c_float), d_, K,
                                                 _p(x, C.c_float), xn, h, _p(out, C.c_float))
    if rc != 0:
        raise RuntimeError("psh_oracle_all_distances_embedded failed")
    return out


def gather_paths(dataset, idx, length: int, r_offset: int = 0) -> np.ndarray:
    ds = _rows(dataset)
    ix = np.ascontiguousarray(idx, dtype=np.int32)
    n = ix.size // 2
    out = np.empty(ix.shape[:-1] + (length,), np.float32)
    rc = lib().psh_oracle_gather_paths(_p(ds, C.c_float), ds.shape[0], ds.shape[1], r_offset,
                                       _p(ix, C.c_int32), n, length, _p(out, C.c_float))
    if rc != 0:
        raise RuntimeError("psh_oracle_gather_paths: index out of range")
    return out


def shadow(dataset, x_context, k: int, horizon=None, nthreads: int = 0):
    """Whole shadow() result of path_shadowing.py:181-218 for the Identity +
    RelativeMSE + PredictionContext(horizon) configuration:
    (d (B,k), paths (B,k,1,W+h), idx (B,k,2))."""
    h = 0 if horizon is None else int(horizon)
    q = np.atleast_2d(_f32(x_context).reshape(-1, np.asarray(x_context).shape[-1]))
    d, idx = scan_topk(dataset, q, k, h=h, nthreads=nthreads)
    paths = gather_paths(dataset, idx, q.shape[1] + h)[:, :, None, :]
    return d, paths, idx
