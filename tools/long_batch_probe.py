"""Long Identity windows (34 <= W <= 256) with SEVERAL queries: the one call (whatever psh_scan_topk routes it to) beside a
host-side loop of one-query calls (the matrix-core long-window scan), configs[1]'s ensemble.  One JSON line per (W, B)."""
import argparse
import json
import time

import torch

from shadowing_amd import _native as N


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
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    ds = torch.randn((args.R, args.T), generator=g).cumsum(1).mul_(0.05).to(dev)
    ws = N.Workspace(dev)
    for W in args.W:
        for B in args.B:
            rows = torch.randint(0, args.R, (B,), generator=g)
            q = torch.stack([ds[int(r), 100:100 + W] for r in rows]).contiguous() + 0.01 * torch.randn((B, W), generator=g).to(dev)
            info = {}
            d0, i0, st0 = N.scan_topk(ds, q, args.k, h=0, workspace=ws, info=info)
            one = lambda: N.scan_topk(ds, q, args.k, h=0, workspace=ws)
            parts = [q[b:b + 1].contiguous() for b in range(B)]
            loop = lambda: [N.scan_topk(ds, p, args.k, h=0, workspace=ws) for p in parts]
            res = loop()
            torch.cuda.synchronize()
            ok = bool((st0 == 0).all()) and all(bool((r[2] == 0).all()) for r in res)
            same = ok and all(torch.equal(res[b][0][0], d0[b]) and torch.equal(res[b][1][0], i0[b]) for b in range(B))
            print(json.dumps(dict(W=W, B=B, path=info.get("path"), call_ms=round(timed(one, args.steps), 3),
                                  loop_ms=round(timed(loop, args.steps), 3), status_ok=ok, same=same)), flush=True)


if __name__ == "__main__":
    main()
