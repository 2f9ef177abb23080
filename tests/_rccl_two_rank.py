"""Two (or more) real ranks over RCCL, launched by tests/test_gpu_configs3.py::test_two_rank_rccl_run through
torch.distributed.run wherever that many GPUs are visible.  Every rank owns a contiguous block of rows, runs the collective
scan (library exchange AND torch exchange, single query and a batch, serial and pipelined) and compares with the oracle's scan
of the WHOLE ensemble.  Prints RANK-OK per rank."""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import oracle
    import shadowing_amd as sa
    from _util import assert_exact
    from shadowing_amd import synthetic as syn
    from shadowing_amd.distributed import ShardedPathShadowing
    oracle.build()
    R_per, T, W, h, k = 8192, 4096, 20, 20, 1024
    whole = np.concatenate([syn.dataset(R_per, T, seed=100 + g) for g in range(world)], 0)
    mine = whole[rank * R_per:(rank + 1) * R_per]
    try:
        for exchange in ("library", "torch"):
            obj = ShardedPathShadowing(sa.Identity(W), sa.RelativeMSE(), mine, rank * R_per, sa.PredictionContext(h), device=dev,
                                       exchange=exchange)
            for B in (1, 6):
                q = syn.rolling_queries(B, W, 200 + B)
                d, paths, idx = obj.shadow(q, k)
                od, opaths, oidx = oracle.shadow(whole, q, k, h)
                assert_exact(d, idx, od, oidx, f"rank {rank} {exchange} B={B}")
                assert np.array_equal(paths, opaths)
            # independent single queries, pipelined: the exchange of query i beside the scan of query i + 1
            qs = [torch.tensor(syn.gbm_log_returns((1, W), 300 + i)) for i in range(6)]
            outs, pend = [], None
            for qi in qs:
                nxt = obj.scan_begin(qi, k)
                if pend is not None:
                    outs.append(pend.finish())
                pend = nxt
            outs.append(pend.finish())
            torch.cuda.synchronize()
            for qi, (dd, ii) in zip(qs, outs):
                od, oi = oracle.scan_topk(whole, qi.numpy(), k, h=h)
                assert_exact(dd.cpu().numpy(), ii.cpu().numpy(), od, oi, f"rank {rank} {exchange} pipelined")
            obj.close()
        print("RANK-OK", rank, flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
