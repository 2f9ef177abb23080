#!/usr/bin/env python
"""Host-side cost of the sharded step (one rank, RCCL process group of size 1): serial scan() against the
pipelined scan_begin()/finish() loop bench.py uses -- wall time per step and the time the Python loop itself
takes to enqueue a step (if that exceeds the GPU time per step, the loop is host-bound)."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import shadowing_amd as sa  # noqa: E402
from shadowing_amd import synthetic as syn  # noqa: E402
from shadowing_amd.distributed import ShardedPathShadowing  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
R, T, W, h, k = 32768, 4096, 20, 20, 1024
ds = torch.from_numpy(syn.dataset(R, T, 0)).to(dev)
q = torch.from_numpy(syn.single_query(W, syn.QUERY_SEED)[None, :]).to(dev)
obj = ShardedPathShadowing(sa.Identity(W), sa.RelativeMSE(), ds, 0, sa.PredictionContext(h), device=dev, always_exchange=True)
N = 300


def serial():
    for _ in range(N):
        obj.scan(q, k, check=False)


def pipelined():
    pend = None
    for _ in range(N):
        nxt = obj.scan_begin(q, k, check=False)
        if pend is not None:
            pend.finish()
        pend = nxt
    pend.finish()


for name, fn in (("serial", serial), ("pipelined", pipelined), ("serial", serial), ("pipelined", pipelined)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:10s} wall {1e6 * (t2 - t0) / N:7.1f} us/step   enqueue {1e6 * (t1 - t0) / N:7.1f} us/step")
dist.destroy_process_group()
