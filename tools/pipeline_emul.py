#!/usr/bin/env python
"""One-GPU emulation of the sharded step at G = 8 (no collective available here): scan of the local shard, then an
"exchange" (the local list copied into 8 slots of a gather buffer, rows offset so that the lists stay ordered) and the
G = 8 sorted-list merge.  Compares where the exchange + merge of step i is enqueued:
  serial    : on the compute stream, in front of the next scan
  behind    : on the compute stream, behind the next scan (what ShardedPathShadowing's begin/finish pipelining does)
  side      : on a second stream, concurrently with the bootstrap / threshold kernels of the next scan
Prints us per step for each."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from shadowing_amd import _native, synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
R, T, W, h, k, G = 32768, 4096, 20, 20, 1024, 8
ds = torch.from_numpy(syn.dataset(R, T, 0)[:, 0, :].copy()).to(dev)
q = torch.from_numpy(syn.single_query(W, syn.QUERY_SEED)[None, :]).to(dev)
ws = _native.Workspace(dev)
side = torch.cuda.Stream(device=dev)
N = 300


def scan():
    send = torch.empty(3 * k, dtype=torch.int32, device=dev)
    out = (send[:k].view(torch.float32).view(1, k), send[k:].view(1, k, 2))
    _native.scan_topk(ds, q, k, h=h, workspace=ws, out=out)
    return send


def exchange_and_merge(send):
    gathered = send.repeat(G, 1)                                   # stands in for the all-gather's receive buffer
    gathered[:, k::2] += (torch.arange(G, device=dev, dtype=torch.int32) * R)[:, None]   # rank g owns rows g*R ...
    return _native.merge_sorted_gathered(gathered, G, 1, k, k)


def run(mode):
    pend = None
    for _ in range(N):
        send = scan()
        if mode == "serial":
            exchange_and_merge(send)
        elif mode == "behind":
            if pend is not None:
                exchange_and_merge(pend)
            pend = send
        else:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if DELAY_CYCLES:
                    torch.cuda._sleep(DELAY_CYCLES)                # stands in for the collective's latency
                exchange_and_merge(send)
                done = torch.cuda.Event()
                done.record()
            send.record_stream(side)
            pend = done
    if mode == "behind" and pend is not None:
        exchange_and_merge(pend)
    if mode == "side" and pend is not None:
        torch.cuda.current_stream().wait_event(pend)


DELAY_CYCLES = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
torch.cuda._sleep(10_000_000)
torch.cuda.synchronize()
TICKS_PER_US = 10_000_000 / ((time.perf_counter() - t0) * 1e6)      # calibrate torch.cuda._sleep's unit
print(f"_sleep: {TICKS_PER_US:.1f} ticks per us")
for mode in ("serial", "behind", "side"):
    run(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(mode)
    torch.cuda.synchronize()
    print(f"{mode:7s} {1e6 * (time.perf_counter() - t0) / N:7.1f} us/step")
for us in (10, 20, 30, 45):
    DELAY_CYCLES = int(us * TICKS_PER_US)
    run("side")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run("side")
    torch.cuda.synchronize()
    print(f"side + {us:2d} us of collective latency: {1e6 * (time.perf_counter() - t0) / N:7.1f} us/step")
