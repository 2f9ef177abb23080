"""Identity windows with SEVERAL queries: the call (whatever psh_scan_topk routes it to -- 34 <= W <= 256: a loop of one-query
matrix-core steps inside the call), the same call with PSH_FLAG_FILTER_VALU (one pass, vector-ALU filter) and a host-side
loop of one-query calls, on configs[1]'s ensemble (--walk: price levels, the bounds' worst case).  One JSON line per (W, B)."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch

from shadowing_amd import _native as N
from shadowing_amd import synthetic as syn


def timed(fn, steps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=32768)
    ap.add_argument("--T", type=int, default=4096)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--W", type=int, nargs="+", default=[64, 126, 252])
    ap.add_argument("--B", type=int, nargs="+", default=[1, 2, 4, 16, 64])
    ap.add_argument("--walk", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    if args.walk:      # price LEVELS (smooth): the f16 / 8-bit bounds' slack dwarfs the admission level -- the worst case
        ds = torch.randn((args.R, args.T), generator=g).cumsum(1).mul_(0.05).to(dev)
    else:              # the benchmark's ensemble (log-returns)
        ds = torch.as_tensor(syn.dataset(args.R, args.T, 2024)[:, 0, :].copy()).to(dev)
    ws = N.Workspace(dev)
    for W in args.W:
        for B in args.B:
            if args.walk:
                q = torch.randn((B, W), generator=g).cumsum(1).mul_(0.05).to(dev)
            else:
                q = torch.as_tensor(syn.rolling_queries(B, W, 2025)).to(dev)
            info = {}
            d0, i0, st0 = N.scan_topk(ds, q, args.k, h=0, workspace=ws, info=info)
            one = lambda: N.scan_topk(ds, q, args.k, h=0, workspace=ws)
            valu = lambda: N.scan_topk(ds, q, args.k, h=0, workspace=ws, flags=N.FLAG_FILTER_VALU)
            parts = [q[b:b + 1].contiguous() for b in range(B)]
            loop = lambda: [N.scan_topk(ds, p, args.k, h=0, workspace=ws) for p in parts]
            res = loop()
            torch.cuda.synchronize()
            ok = bool((st0 == 0).all()) and all(bool((r[2] == 0).all()) for r in res)
            same = ok and all(torch.equal(res[b][0][0], d0[b]) and torch.equal(res[b][1][0], i0[b]) for b in range(B))
            print(json.dumps(dict(data="walk" if args.walk else "returns", W=W, B=B, path=info.get("path"), call_ms=round(timed(one, args.steps), 3), one_pass_valu_ms=round(timed(valu, args.steps), 3),
                                  loop_ms=round(timed(loop, args.steps), 3), status_ok=ok, same=same)), flush=True)


if __name__ == "__main__":
    main()
