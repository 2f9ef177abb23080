#!/usr/bin/env python
"""bench.py -- windows scanned/sec of the k-nearest-path scan on N MI355X.

A "step" is ONE pass of the hot path over one batch: the whole
PathShadowing scan (query prep -> sample -> threshold -> sliding-window scan ->
select; for N > 1 also the RCCL all-gather of the per-shard top-k and the merge)
with the trajectory ensemble already resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: spawns its N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --sweep 1,2,4,8                      (one line per N, weak-scaling efficiency on stderr)

Workload (BASELINE.json configs[1], the one the metric is quoted on):
  R = 32768 trajectories per GPU x T = 4096, W = 20, horizon 20, k = 1024, one query;
  synthetic GBM log-returns (shadowing_amd.synthetic), weak scaling R_total = N * 32768.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     the dominant kernel (sliding-window scan) against the 8 TB/s HBM peak:
               algorithmic bytes per launch / its average duration, measured live with
               HIP events recorded around that kernel on its own stream, inside the
               timed loop;
  cpu_baseline the CPU oracle (a port of the reference's algorithm, oracle/) timed on
               this box's host cores on the same workload -- a reported baseline, not
               a target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

# the host driver of these boxes only supports dmabuf IPC: without this RCCL fails with `hipIpcGetMemHandle: invalid argument`
# (exported on the GPU boxes already; set here too, before the HIP runtime starts, for whoever launches the ranks)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBPS = 6290.0   # same guide: what a float4 copy kernel measures on this part (79 % of the spec) -- reported beside `frac`


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000,
                    help="timed steps (default 1000 = ~0.1 s of device time: the first ~3 ms of load after an idle period are "
                         "slower on these boxes for every kernel -- tools/cold_probe.py -- and a long run averages that out)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows-per-gpu", type=int, default=32768)
    ap.add_argument("--T", type=int, default=4096)
    ap.add_argument("--W", type=int, default=20)
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--queries", type=int, default=1, help="B: 1 = configs[1]; 512 = configs[2] (rolling windows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the golden-vector check of the first result")
    ap.add_argument("--force-sharded", action="store_true",
                    help="exercise the multi-GPU code path (process group, all-gather, merge) even with one rank")
    ap.add_argument("--filter", choices=("mx", "valu"), default="mx",
                    help="rejection test of the scan: matrix cores (default) or vector ALUs (PSH_FLAG_FILTER_VALU; comparison runs)")
    ap.add_argument("--no-fuse", action="store_true", help="the separate bootstrap / threshold / scan / select launches")
    ap.add_argument("--mq-f16", action="store_true",
                    help="--queries > 1: the batched scan's rejection test as the f16 product (PSH_FLAG_MQ_F16) instead of the 8-bit one (A/B)")
    ap.add_argument("--streams", type=int, default=0,
                    help="single query: consecutive steps (independent queries) are issued round-robin on this many HIP "
                         "streams as the three overlap-friendly launches of PSH_FLAG_OVERLAP (sample + level, barrier-free "
                         "scan, ranking), so one step's latency-bound launches run beside another step's scan; 1 = one "
                         "stream, the fused single launch; 0 (default) = 3, or 2 for the row-sharded step, whose exchange "
                         "and merge run on a stream of their own (measured on one rank with the exchange forced: 88.7 us "
                         "per step on 2 scan streams, 105.5 on 3, 114.5 on 1)")
    ap.add_argument("--cpu-oracle", action="store_true",
                    help="TEST HOOK, measures nothing: the row-sharded control flow of this file (process group, step / drain / "
                         "timed_region, the parity checks, the JSON line) on CPU tensors over gloo, with the CPU oracle standing in "
                         "for the HIP scan -- tests/test_bench_flow_cpu.py runs it with two ranks so that a Python-level bug "
                         "cannot burn a multi-GPU lease")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (warm-up + K steps) is run this many times: `value` / `ms_per_step` are the FIRST region's (the "
                         "contract's K steps), `ms_per_step_repeats` carries median / min / max over all of them")
    ap.add_argument("--no-blocking-api", action="store_true",
                    help="skip the `blocking_shadow_api` block (N = 1, one query): the reference's own call -- one blocking "
                         "PathShadowing.shadow(x, k, cuda=True), README.md:47-57 -- timed call by call outside the headline's region")
    ap.add_argument("--sweep", type=str, default=None,
                    help="comma-separated GPU counts: run each in turn, print one JSON line per N (stdout) and the "
                         "weak-scaling efficiency against the first (stderr)")
    return ap.parse_args()


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` without a launcher: run the N ranks through torch.distributed.run (one process per
    GPU, RCCL over xGMI), exactly as the driver's own command line does.  Rank 0's JSON line passes through."""
    import subprocess
    if "--cpu-oracle" not in argv:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            # ONE clear line instead of n tracebacks from the ranks
            sys.stderr.write(f"bench.py: --gpus {n} needs {n} visible GPUs, this node shows {have} (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)\n")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = 1
    for attempt in range(3):
        # the rendezvous port is picked by binding port 0 and closing it again: another process can take it in between --
        # a rendezvous that dies of EADDRINUSE is started again on a fresh port (the ranks have not touched a GPU by then)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), str(Path(__file__).resolve())] + argv
        rc, head, seconds = _run_streaming(cmd, env)
        # only a rendezvous that died EARLY of a taken port is started again: a run that got as far as its GPUs is not repeated
        busy = rc != 0 and seconds < 60.0 and any(m in head for m in ("EADDRINUSE", "Address already in use", "address already in use"))
        if not busy:
            break
        sys.stderr.write(f"bench.py: rendezvous port was taken (attempt {attempt + 1}); retrying on another one\n")
    return rc


def _run_streaming(cmd: list[str], env: dict) -> tuple[int, str, float]:
    """Run `cmd`, passing its stderr through LINE BY LINE as it comes (progress, RCCL / HIP errors of the ranks are seen while the
    run is alive, nothing is held back in memory) and keeping the first 64 KB of it for the caller to look at.  Returns
    (return code, that head of stderr, wall seconds)."""
    import subprocess
    import threading
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True)
    head: list[str] = []
    kept = [0]

    def pump():
        for line in proc.stderr:
            sys.stderr.write(line)
            sys.stderr.flush()
            if kept[0] < 65536:
                head.append(line)
                kept[0] += len(line)

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    rc = proc.wait()
    th.join(timeout=10)
    return rc, "".join(head), time.perf_counter() - t0


def sweep(counts: list[int], argv: list[str]) -> int:
    """N = counts[0], counts[1], ... back to back; stdout: one JSON line per N; stderr: ms/step and weak-scaling
    efficiency (value_N / (N/N0 * value_N0)) per N."""
    import subprocess
    rest = []
    skip = False
    for a in argv:                                       # drop --sweep X / --gpus X from the forwarded arguments
        if skip:
            skip = False
        elif a in ("--sweep", "--gpus"):
            skip = True
        elif not a.startswith(("--sweep=", "--gpus=")):
            rest.append(a)
    lines = []
    for n in counts:
        res = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--gpus", str(n)] + rest, capture_output=True, text=True)
        line = next((ln for ln in res.stdout.splitlines() if ln.startswith("{")), None)
        if res.returncode != 0 or line is None:
            sys.stderr.write(f"[sweep] N={n}: failed (rc {res.returncode})\n{res.stderr[-2000:]}\n")
            return 1
        print(line, flush=True)
        lines.append(json.loads(line))
    for row in sweep_report(lines):
        sys.stderr.write(row + "\n")
    return 0


def sweep_report(lines: list[dict]) -> list[str]:
    """One text row per JSON line of a sweep: weak-scaling efficiency = value_N / (N / N0 * value_N0) against the FIRST
    line.  A line whose run did not verify itself (parity false, or a multi-rank line whose RCCL world is not its
    n_gpus) is marked: its number is not evidence."""
    base = lines[0]
    rows = []
    for j in lines:
        eff = j["value"] / (base["value"] * j["n_gpus"] / base["n_gpus"])
        marks = []
        if j["n_gpus"] > 1 and j.get("rccl_world_size") != j["n_gpus"]:
            marks.append("RCCL-WORLD-MISMATCH")
        if j.get("parity_vs_golden") is False or (j.get("parity_rotating_queries") or {}).get("ok") is False:
            marks.append("PARITY-FAILED")
        if j.get("parity_vs_golden") is None:
            marks.append("parity-unchecked")
        rows.append(f"[sweep] n_gpus={j['n_gpus']} rccl_world_size={j.get('rccl_world_size')} ms_per_step={j['ms_per_step']:.4f} "
                    f"value={j['value']:.4g} {j['unit']} weak_scaling_efficiency={eff:.3f}" + ("".join(" " + m for m in marks)))
    return rows


def workload_label(R: int, T: int, W: int, h: int, k: int, B: int, world: int) -> str:
    """config.workload: which BASELINE.json configuration this run IS, derived from the sizes and the rank count actually
    used -- never a fixed string (a run with other --rows-per-gpu / --gpus must not carry configs[1]'s name)."""
    sizes = f"R={R} paths/GPU x T={T}, W={W}, horizon={h}, k={k}"
    base = (T, W, h, k) == (4096, 20, 20, 1024)
    if B == 1 and base and R == 32768:
        if world == 1:
            return "BASELINE.json configs[1]: single query, " + sizes + ", 1 GPU"
        if world == 8:
            return f"BASELINE.json configs[3]: R={world * R} paths sharded 8 ways (local top-k + one RCCL all-gather + merge), single query, " + sizes
        return (f"BASELINE.json configs[3]'s layout at {world} of its 8 GPUs (weak-scaling point): R={world * R} paths sharded {world} ways, "
                "single query, " + sizes)
    if B == 512 and base and R == 32768 and world == 1:
        return "BASELINE.json configs[2]: 512 batched query dates (rolling window), " + sizes + ", 1 GPU"
    what = "single query" if B == 1 else f"batched queries B={B} (rolling window)"
    return f"not a BASELINE.json configuration (non-default sizes): {what}, {sizes}, {world} GPU" + ("s" if world > 1 else "") + \
           (f", R_total={world * R} sharded {world} ways" if world > 1 else "")


def host_merge(d_all: np.ndarray, i_all: np.ndarray, k: int):
    """k best by (d, r, t) out of (B, n) candidates on the host (numpy lexsort); entries with r < 0 are padding.  The
    checker's merge: the distributed oracle of the parity checks and the --cpu-oracle hook use it, never a timed step."""
    B = d_all.shape[0]
    out_d = np.empty((B, k), np.float32)
    out_i = np.empty((B, k, 2), np.int32)
    for b in range(B):
        real = i_all[b, :, 0] >= 0
        dd, ii = d_all[b][real], i_all[b][real]
        o = np.lexsort((ii[:, 1], ii[:, 0], dd))[:k]
        out_d[b], out_i[b] = dd[o], ii[o]
    return out_d, out_i


def same_result(d: np.ndarray, idx: np.ndarray, ed: np.ndarray, eidx: np.ndarray, tie_free_order: bool) -> bool:
    """Bit-equal distances and identical (r, t).  `tie_free_order` False: the expected rows come from the reference, whose
    order among exactly tied distances is arbitrary (unstable partial sort): sorted distances and the SET of pairs."""
    if tie_free_order:
        return bool(np.array_equal(d.view(np.uint32), ed.view(np.uint32)) and np.array_equal(idx, eidx))
    if not np.array_equal(d.view(np.uint32), np.sort(ed, axis=1).view(np.uint32)):
        return False
    return all({tuple(v) for v in idx[b]} == {tuple(v) for v in eidx[b]} for b in range(idx.shape[0]))


def cpu_baseline(ds: np.ndarray, q: np.ndarray, k: int, h: int) -> dict:
    """The oracle (checker) timed as the CPU baseline: all host cores, a bounded sample of
    the same workload (about 10-20 s of CPU work)."""
    import oracle
    oracle.build()
    cores = os.cpu_count() or 1
    R, _, T = ds.shape
    W = q.shape[-1]
    nq = 1
    rows = R                                              # the whole workload, repeated for ~10 s
    oracle.scan_topk(ds[:rows], q[:nq], k, h=h, nthreads=cores)       # warm-up
    times = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        oracle.scan_topk(ds[:rows], q[:nq], k, h=h, nthreads=cores)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > 10.0 or len(times) >= 200:
            break
    best, med = min(times), sorted(times)[len(times) // 2]
    windows = rows * (T - W - h + 1) * nq
    out = {"value": windows / med, "unit": "windows/s", "cores": cores, "kind": "port",
           "sample": f"{rows} of {R} rows x {nq} query of the same workload, {len(times)} passes in "
                     f"{sum(times):.1f} s, median {med:.3f} s (best {best:.3f} s), oracle/psh_oracle.c, "
                     f"OpenMP over rows, gcc -O3 -mavx2 -mfma"}
    out["torch_formulation"] = torch_cpu_baseline(ds, q, k, h)
    return out


def torch_cpu_baseline(ds: np.ndarray, q: np.ndarray, k: int, h: int) -> dict:
    """SURVEY 8d's second CPU figure: the torch expression of the same math -- what the reference executes with
    cuda=False (conv1d with the one-hot kernel, broadcast norm, topk, running merge; here the package's own
    generic path, PathShadowing._generic_scan) -- on a bounded row sample, all of torch's intra-op threads."""
    import shadowing_amd as sa
    R, _, T = ds.shape
    W = q.shape[-1]
    rows = min(R, 2048)
    obj = sa.PathShadowing(sa.Identity(W), sa.RelativeMSE(), ds[:rows], sa.PredictionContext(h))
    x = torch.tensor(q[:1])[:, None, :]
    y = torch.as_tensor(ds[:rows])
    kk = min(k, rows * (T - W - h + 1))
    obj._generic_scan(x, y, kk, 16, False)                 # warm-up
    times = []
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < 5.0 and len(times) < 20:
        t0 = time.perf_counter()
        obj._generic_scan(x, y, kk, 16, False)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(rows * (T - W - h + 1) / med, 1), "unit": "windows/s", "threads": torch.get_num_threads(),
            "sample": f"{rows} of {R} rows x 1 query, 16 splits, {len(times)} passes, median {med:.3f} s; "
                      f"shadowing_amd.PathShadowing._generic_scan (the reference's formulation in torch ops), cuda=False"}


def blocking_shadow_api(sa, syn, ds, ds_host, q_hosts, W, h, k, expected_by_oracle):
    """The reference's OWN call, timed call by call: one blocking `PathShadowing.shadow(x, k, cuda=True)` on a resident
    ensemble, numpy query in, numpy (distances, paths, indices) out (reference README.md:47-57, path_shadowing.py:181-218).
    Outside the headline's timed region; wall clock of the caller (time.perf_counter around every call)."""
    obj = sa.PathShadowing(sa.Identity(W), sa.RelativeMSE(), ds, sa.PredictionContext(h))
    unrelated = [syn.single_query(W, 5000 + j) for j in range(64)]
    WARM, CALLS = 300, 500

    def loop(o, queries, n):
        ts = []
        for i in range(n):
            q = queries[i % len(queries)]
            t0 = time.perf_counter()
            o.shadow(q, k=k, cuda=True)
            ts.append(1e6 * (time.perf_counter() - t0))
        return np.asarray(ts)

    loop(obj, unrelated, WARM)
    ts = loop(obj, unrelated, CALLS)
    # parity: the rotating queries of the headline through the SAME call, paths included (the oracle's distances and
    # indices; the paths are the ensemble's own samples at those indices)
    parity = None
    if expected_by_oracle is not None:
        parity = True
        for qi, qh in enumerate(q_hosts):
            d, paths, idx = obj.shadow(qh[0], k=k, cuda=True)
            od, oi = expected_by_oracle(qi)
            want = np.stack([ds_host[r, :, t:t + W + h] for r, t in oi[0]])[None]
            parity = parity and same_result(d, idx, od, oi, tie_free_order=True) and bool(np.array_equal(paths, want))
        if not parity:
            raise SystemExit("PARITY FAILURE of the blocking shadow() calls against the oracle")
    served = "psh_shadow_blocking: " + ("psh::scan_fused_kernel<..,BLK> (scan + selection + path gather in one launch, results written "
                                        "to pinned host memory by the kernel, completion words polled)" if getattr(obj, "_sync_slot", None) and obj._sync_slot[1].last_fused
                                        else "psh_scan_topk's launches + the gather launch, stream synchronised")
    out = {"what": "one blocking PathShadowing.shadow(x, k, cuda=True) per call on a resident ensemble -- the reference's own call "
                   "(README.md:47-57) -- numpy in, numpy (d, paths, idx) out; caller's wall clock per call, unrelated queries",
           "median_us": round(float(np.median(ts)), 1), "p90_us": round(float(np.percentile(ts, 90)), 1),
           "min_us": round(float(ts.min()), 1), "max_us": round(float(ts.max()), 1), "calls": CALLS, "warmup_calls": WARM,
           "parity_vs_oracle": parity, "served_by": served}
    # rolling query dates (what predict() loops over, ref :286-301): without hints and with hint="auto" -- the previous date's
    # k-th distance as this date's admission level, a hint a caller CAN have
    rolling = list(syn.rolling_queries(CALLS, W, 7))
    loop(obj, rolling, 50)
    t_plain = loop(obj, rolling, CALLS)
    hinted = sa.PathShadowing(sa.Identity(W), sa.RelativeMSE(), ds, sa.PredictionContext(h), hint="auto")
    loop(hinted, rolling, 50)
    seen = {"ok": 0, "short": 0, None: 0}
    t_hint = []
    for q in rolling:
        t0 = time.perf_counter()
        hinted.shadow(q, k=k, cuda=True)
        t_hint.append(1e6 * (time.perf_counter() - t0))
        seen[hinted.last_hint] += 1
    out["rolling_dates"] = {"no_hint_median_us": round(float(np.median(t_plain)), 1), "hint_auto_median_us": round(float(np.median(t_hint)), 1),
                            "no_hint_mean_us": round(float(np.mean(t_plain)), 1), "hint_auto_mean_us": round(float(np.mean(t_hint)), 1),
                            "hints_held": seen["ok"], "hints_fell_short": seen["short"], "calls_without_hint": seen[None], "calls": CALLS,
                            "policy": "PathShadowing(hint=\"auto\"): level = (previous date's d_k x ||x||)^2 x min(1.15, 3^(2/W)); a hint "
                                      "that falls short costs one more launch and switches hints off for a few calls"}
    if k != 8192:
        big = 8192
        for q in unrelated[:20]:
            obj.shadow(q, k=big, cuda=True)
        tb = loop_k(obj, unrelated, 100, big)
        out["k8192_median_us"] = round(float(np.median(tb)), 1)
    return out


def loop_k(o, queries, n, k):
    ts = []
    for i in range(n):
        q = queries[i % len(queries)]
        t0 = time.perf_counter()
        o.shadow(q, k=k, cuda=True)
        ts.append(1e6 * (time.perf_counter() - t0))
    return np.asarray(ts)


def main():
    args = parse()
    if args.sweep:
        raise SystemExit(sweep([int(c) for c in args.sweep.split(",")], sys.argv[1:]))
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    # stdout carries ONE line, the JSON: anything a native library prints there (RCCL writes a five-line version banner
    # to stdout when a communicator goes away) is sent to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    on_gpu = not args.cpu_oracle
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs a HIP device"
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    import torch.distributed as dist
    use_pg = world > 1 or args.force_sharded or not on_gpu
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import shadowing_amd as sa
    from shadowing_amd import _native, synthetic as syn
    from shadowing_amd.distributed import ShardedPathShadowing
    if on_gpu:
        _native.load()   # fail loudly if the HIP extension is missing

    R, T, W, h, k, B = args.rows_per_gpu, args.T, args.W, args.horizon, args.k, args.queries
    Tp = T - W - h + 1
    flags = (_native.FLAG_FILTER_VALU if args.filter == "valu" else 0) | (_native.FLAG_NO_FUSE if args.no_fuse else 0)
    if args.mq_f16:
        flags |= _native.FLAG_MQ_F16
    # independent single queries on several streams: the overlap-friendly launches (the library falls back to the fused /
    # separate launches by itself where they do not apply)
    want_streams = args.streams if args.streams > 0 else (2 if use_pg else 3)
    n_streams = want_streams if (B == 1 and not args.no_fuse and args.filter != "valu" and want_streams > 1) else 1
    if n_streams > 1:
        flags |= _native.FLAG_OVERLAP
    # per-rank block of the ensemble: block g is dataset(R, T, seed=g); rank 0's block at
    # the default sizes is exactly the dataset of tests/golden/cfg2_R32768.npz
    ds_host = syn.dataset(R, T, seed=rank)
    # NQ DISTINCT query batches take turns through the steps (step c: batch c % NQ on stream c % n_streams): a result left
    # in a stream's buffers at the end of the timed region is checked against what ITS query must give, so workspace or
    # buffer aliasing between co-resident steps cannot go unnoticed at speed.  Batch 0 is the golden fixtures' query.
    NQ = 4
    q_seeds = [syn.QUERY_SEED] + [1000 + j for j in range(1, NQ)]
    q_hosts = [np.ascontiguousarray(syn.single_query(W, sd)[None, :] if B == 1 else syn.rolling_queries(B, W, sd)) for sd in q_seeds]
    q_host = q_hosts[0]
    if not on_gpu:
        n_streams = 1
    ds = torch.from_numpy(ds_host).to(dev)                    # resident in HBM before any timing
    qs = [torch.from_numpy(qh).to(dev) for qh in q_hosts]
    q = qs[0]
    ws = _native.Workspace(dev) if on_gpu else None
    # one workspace per stream (its header carries per-call state), results of a stream's steps in its own buffers
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)] if n_streams > 1 else [None]
    wss = [ws] + [_native.Workspace(dev) for _ in range(n_streams - 1)]

    def oracle_local(ds2d, qq, kk, hh, r_offset):
        # (--cpu-oracle only) the checker stands in for the HIP scan so that the control flow runs without a GPU
        import oracle
        d_, i_ = oracle.scan_topk(ds2d.numpy(), qq.numpy(), kk, h=hh, r_offset=r_offset, nthreads=2)
        return torch.from_numpy(d_), torch.from_numpy(i_)

    def torch_host_merge(d_all, i_all, kk):
        d_, i_ = host_merge(d_all.numpy(), i_all.numpy(), kk)
        return torch.from_numpy(d_), torch.from_numpy(i_)

    sharded = None
    if use_pg and on_gpu:
        sharded = ShardedPathShadowing(sa.Identity(W), sa.RelativeMSE(), ds, rank * R, sa.PredictionContext(h), device=dev,
                                       always_exchange=args.force_sharded, streams=n_streams)
    elif use_pg:
        sharded = ShardedPathShadowing(sa.Identity(W), sa.RelativeMSE(), ds, rank * R, sa.PredictionContext(h),
                                       local_topk=oracle_local, merge=torch_host_merge, always_exchange=True)

    # HIP events around the dominant kernel on every EV_EVERY-th timed step: an event record is a barrier packet of its own
    # on the stream (~5 us for the pair, measured: 112 vs 101 us per step with a pair on every step), so bracketing every
    # launch would tax the very step time the metric is
    EV_EVERY = 4
    ev_pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                for _ in range((args.steps + EV_EVERY - 1) // EV_EVERY)] if on_gpu else []
    for a, b in ev_pairs:   # materialise the hipEvent handles
        a.record(); b.record()
    sync()

    statuses = []
    last_query = {}    # stream -> the query batch of the last step issued on it
    step_flags = [flags]
    cfg = {"n_streams": n_streams, "count": 0}
    outs = [(torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B, k, 2), dtype=torch.int32, device=dev))
            for _ in range(n_streams)]
    status_ring = list(torch.zeros((args.steps + args.warmup + 8, B), dtype=torch.int32, device=dev).unbind(0))
    sync()

    hints = [None] * NQ  # single query: per rotating query the caller-side admission level of the hinted one-stream comparison
    pending = []       # sharded path: (the step whose all-gather is still in flight, its query batch)
    from collections import deque
    finished = deque(maxlen=2)     # sharded path: the last merged results (they live in the ring: valid until 2 x streams further steps)

    def step(i=None):
        c = cfg["count"]
        cfg["count"] = c + 1
        qi = c % NQ
        q = qs[qi]
        if sharded is not None:
            # steps are independent query batches: step i+1 begins (local scan, start of its all-gather) before step i
            # is finished (wait for its all-gather on the compute stream, merge) -- the ~20 us collective latency runs
            # under the next scan.  Every step's merged result is complete when the timed region ends (drain()).
            nxt = sharded.scan_begin(q, k, check=False, queries_ready=True)   # no host sync inside the timed loop; the queries were uploaded long ago
            if sharded.last_status is not None and not sharded._fast:
                statuses.append(sharded.last_status)       # (the prepared ring keeps every step's status word itself: status_max())
            out = None
            if pending:
                pend, pq = pending.pop()
                out = pend.finish()
                finished.append((pq, out))
            pending.append((nxt, qi))
            return out
        ev = ev_pairs[i // EV_EVERY] if (i is not None and i % EV_EVERY == 0) else None
        si = c % cfg["n_streams"]
        last_query[si] = qi
        # results go to this stream's own buffers (a later step on the same stream overwrites them, in stream order); the
        # status words of ALL steps are kept: every one of them is looked at after the timed region
        out = (outs[si][0], outs[si][1], status_ring[c % len(status_ring)])
        if streams[si] is None or cfg["n_streams"] == 1:
            d, idx, st = _native.scan_topk(ds[:, 0, :], q, k, h=h, workspace=wss[0], scan_events=ev, flags=step_flags[0], out=out,
                                           tau_hint=hints[qi] if cfg.get("hint") else None)
        else:
            with torch.cuda.stream(streams[si]):
                d, idx, st = _native.scan_topk(ds[:, 0, :], q, k, h=h, workspace=wss[si], scan_events=ev, flags=step_flags[0], out=out)
        statuses.append(st)
        return d, idx

    # ---- first result: parity against the reference's golden vector (N = 1, default sizes)
    def drain():
        if not pending:
            return None
        pend, pq = pending.pop()
        out = pend.finish()
        finished.append((pq, out))
        return out

    # ---- the checker's answer for a query batch over the WHOLE (sharded) ensemble: every rank runs the CPU oracle on its
    #      own block (its share of the host cores), the lists travel through the process group, rank 0 merges on the host
    expected_cache = {}

    def expected_by_oracle(qi: int, sel=None):
        key = (qi, None if sel is None else tuple(sel))
        if key in expected_cache:
            return expected_cache[key]
        import oracle
        oracle.build()
        qq = q_hosts[qi] if sel is None else q_hosts[qi][list(sel)]
        threads = max(1, (os.cpu_count() or 1) // world)
        od, oi = oracle.scan_topk(ds_host, qq, min(k, R * Tp), h=h, r_offset=rank * R, nthreads=threads)
        if world > 1:
            box = [None] * world
            dist.all_gather_object(box, (od, oi))
            od, oi = host_merge(np.concatenate([b[0] for b in box], axis=1), np.concatenate([b[1] for b in box], axis=1), k)
        expected_cache[key] = (od, oi)
        return od, oi

    parity = None
    first = step()
    d0, i0 = first if sharded is None else drain()
    sync()
    d0, i0 = d0.clone(), i0.clone()
    if statuses and int(statuses[-1].max().item()) != 0:
        raise SystemExit("candidate buffer overflow on the benchmark workload (unexpected)")
    if sharded is not None and sharded._fast and sharded.status_max() != 0:
        raise SystemExit("the first sharded step gave up (status %d; unexpected)" % sharded.status_max())
    default_sizes = (R, T, W, h, k, B) == (32768, 4096, 20, 20, 1024, 1)
    golden_name = None
    if not args.no_parity and default_sizes and world == 1 and (REPO / "tests/golden/cfg2_R32768.npz").exists():
        g = np.load(REPO / "tests/golden/cfg2_R32768.npz")
        if syn.sha256(ds_host) == str(g["dataset_sha256"]):
            golden_name = "tests/golden/cfg2_R32768.npz"
            parity = same_result(d0.cpu().numpy(), i0.cpu().numpy(), g["d"], g["idx"], tie_free_order=False)
            if not parity:
                raise SystemExit("PARITY FAILURE against tests/golden/cfg2_R32768.npz")
    if not args.no_parity and default_sizes and world in (2, 4, 8) and (REPO / "tests/golden/cfg4_R262144.npz").exists():
        # N > 1: the reference's own output on the concatenated rank blocks (tests/golden/make_golden.py --sharded);
        # every rank vouches for its block's bytes, rank 0 compares the first merged result
        g = np.load(REPO / "tests/golden/cfg4_R262144.npz")
        mine = torch.tensor([1.0 if syn.sha256(ds_host) == str(g["block_sha256"][rank]) else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(mine, op=dist.ReduceOp.MIN)
        if float(mine.item()) == 1.0:
            golden_name = f"tests/golden/cfg4_R262144.npz (d_N{world}, idx_N{world})"
            parity = same_result(d0.cpu().numpy(), i0.cpu().numpy(), g[f"d_N{world}"], g[f"idx_N{world}"], tie_free_order=False)
            flag = torch.tensor([1.0 if parity else 0.0], dtype=torch.float64, device=dev)
            dist.broadcast(flag, src=0)
            if float(flag.item()) != 1.0:
                raise SystemExit(f"PARITY FAILURE against {golden_name}")
    parity_first_oracle = None
    if not args.no_parity and parity is None and B == 1 and (world > 1 or not on_gpu):
        # other sizes / rank counts: the first merged result against the distributed oracle
        od, oi = expected_by_oracle(0)
        parity_first_oracle = same_result(d0.cpu().numpy(), i0.cpu().numpy(), od, oi, tie_free_order=True)
        flag = torch.tensor([1.0 if parity_first_oracle else 0.0], dtype=torch.float64, device=dev)
        dist.broadcast(flag, src=0)
        if float(flag.item()) != 1.0:
            raise SystemExit("PARITY FAILURE of the first merged result against the distributed oracle")
    parity_oracle = None
    if world == 1 and on_gpu and not args.no_parity and B > 1:
        # a batch: the first result of the run against the CPU oracle on a subset of the queries spread over the
        # query chunks of the batched scan (first, around the chunk boundary, last)
        import oracle
        oracle.build()
        sel = sorted({0, min(111, B - 1), min(112, B - 1), B // 2, B - 1})
        od, oi = oracle.scan_topk(ds_host, q_host[sel], k, h=h)
        got_d, got_i = d0.cpu().numpy()[sel], i0.cpu().numpy()[sel]
        parity_oracle = bool(np.array_equal(got_d.view(np.uint32), od.view(np.uint32)) and np.array_equal(got_i, oi))
        if not parity_oracle:
            raise SystemExit(f"PARITY FAILURE against the oracle on queries {sel}")

    def timed_region():
        cfg["count"] = 0
        for _ in range(args.warmup):
            step()
        drain()
        statuses.clear()
        sync()
        if sharded is not None:
            sharded.reset_status()
        if use_pg:
            dist.barrier()
        sync()
        cfg["count"] = 0
        finished.clear()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        drain()                                            # the last step's exchange and merge belong to the timed region
        host = time.perf_counter() - t0                    # host time to enqueue all steps (<< elapsed unless host-bound)
        sync()
        if use_pg:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
        bad = int(torch.stack(statuses).max().item()) if statuses else 0
        if sharded is not None:
            bad = max(bad, sharded.status_max())           # EVERY step of the prepared ring, not the slots' last launches
        if use_pg:
            t = torch.tensor([el, float(bad)], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, bad = float(t[0].item()), int(t[1].item())
        return el, host, bad

    # The CU-masked scan streams of the sharded run are BLOCKING streams (hipExtStreamCreateWithCUMask takes no flags):
    # anything enqueued on the legacy default stream would serialise with all of them.  The sharded steps are therefore
    # issued from a stream of their own.
    main_stream = torch.cuda.Stream(dev) if (sharded is not None and n_streams > 1 and on_gpu) else None
    if main_stream is not None:
        main_stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(main_stream)
    elapsed, host_enqueue, bad = timed_region()
    fused_retry = False
    if bad == 2 and not args.no_fuse:
        # a fused launch / an overlap step gave up somewhere (PSH_STATUS_RETRY: its results are invalid): the whole timed
        # region is run again through the separate launches, on every rank, and THAT is what is reported
        fused_retry = True
        flags = (flags & ~_native.FLAG_OVERLAP) | _native.FLAG_NO_FUSE
        step_flags[0] = flags
        if sharded is not None:
            sharded.fuse = False
        elapsed, host_enqueue, bad = timed_region()
    if bad != 0:
        raise SystemExit("candidate buffer overflow during the timed steps (unexpected)")
    # the same region again (the driver's K = 20 steps are 1.7 ms of device time: one region says little about the spread)
    region_ms = [1e3 * elapsed / args.steps]
    for _ in range(max(0, args.repeats - 1)):
        el_r, _, bad_r = timed_region()
        if bad_r != 0:
            raise SystemExit("a repeated timed region reported status %d (unexpected)" % bad_r)
        region_ms.append(1e3 * el_r / args.steps)

    # ---- what the timed steps LEFT BEHIND, against the checker: the last result of every stream (unsharded) / the last two
    #      merged results (sharded) must be what their own query batch gives
    rotating = None
    if not args.no_parity:
        sync()
        left = ([(last_query[si], outs[si]) for si in sorted(last_query)] if sharded is None else list(finished))
        sel = None if B == 1 else sorted({0, min(111, B - 1), min(112, B - 1), B // 2, B - 1})
        ok, checked = True, []
        for qi, (dd, ii) in left:
            od, oi = expected_by_oracle(qi, sel)
            got_d, got_i = dd.cpu().numpy(), ii.cpu().numpy()
            if sel is not None:
                got_d, got_i = got_d[sel], got_i[sel]
            ok = ok and same_result(got_d, got_i, od, oi, tie_free_order=True)
            checked.append(qi)
        if use_pg:
            flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = float(flag.item()) == 1.0
        rotating = {"ok": bool(ok), "query_batches_checked": checked, "distinct_query_batches": NQ,
                    "against": "oracle/psh_oracle.c over the whole ensemble" + (" (every rank its block, host merge)" if world > 1 else ""),
                    "what": "the result each stream's LAST timed step left in its buffers" if sharded is None
                            else "the last two merged results of the timed region"}
        if not ok:
            raise SystemExit(f"PARITY FAILURE after the timed region: results of query batches {checked} differ from the oracle")
    # the bracketed launches of THE timed region (the comparison region below records over the same events)
    scan_ms = [a.elapsed_time(b) for a, b in ev_pairs] if (sharded is None and on_gpu) else []
    end_gaps_ms = [ev_pairs[j][1].elapsed_time(ev_pairs[j + 1][1]) / EV_EVERY for j in range(len(ev_pairs) - 1)] if sharded is None else []
    overlap_mode = bool(flags & _native.FLAG_OVERLAP)
    alg_bytes_const = R * T * 4 + B * W * 4 + B * k * 12    # SURVEY.md 8d (the roofline block below states it again)
    single_stream = None
    if overlap_mode and sharded is None:
        # beside the headline: the same K steps on ONE stream as the fused single launch (what an isolated caller gets)
        keep = (step_flags[0], cfg["n_streams"])
        step_flags[0], cfg["n_streams"] = flags & ~_native.FLAG_OVERLAP, 1
        el1, _, bad1 = timed_region()
        step_flags[0], cfg["n_streams"] = keep
        single_stream = {"ms_per_step": round(1e3 * el1 / args.steps, 5), "value": round(world * R * Tp * B * args.steps / el1, 1),
                         "unit": "windows/s", "launches": (("psh::scan_fused_kernel, one stream" if W <= 33 else "the three launches back to back, one stream")
                                                           if bad1 == 0 else "fused launch gave up (status %d)" % bad1)}
        # ... and the same with the caller's ADMISSION HINT (psh_profile.tau_hint): what a caller that knows the k-th distance
        # of its query roughly gets on one stream -- consecutive rolling dates: the previous date's d_k.  Here every rotating
        # query's hint is its own exact k-th acc x HINT_MARGIN (known from an untimed call), i.e. a 5 % error in d_k: the fused
        # launch then runs no sample phase and no first grid barrier.  Results are checked like the headline's.
        # (The number of windows below a level grows like level^(W / 2) -- W degrees of freedom --: 1.10 on acc admits ~2.6 k at
        #  W = 20 and ~400 k at W = 126.  The same ~2.6 k at every W: margin = 2.6^(2 / W), 1.10 at W = 20, 1.015 at 126.)
        HINT_MARGIN = round(min(1.10, 2.6 ** (2.0 / W)), 4)
        step_flags[0], cfg["n_streams"] = flags & ~_native.FLAG_OVERLAP, 1
        for qi in range(NQ):
            dq, _, stq = _native.scan_topk(ds[:, 0, :], qs[qi], k, h=h, workspace=wss[0], flags=step_flags[0])
            xn = _native.query_norm(qs[qi])
            torch.cuda.synchronize()
            assert int(stq.max().item()) == 0
            hints[qi] = ((dq[:, k - 1].double() * xn.double()) ** 2 * HINT_MARGIN).float().contiguous()
        cfg["hint"] = True
        el2, _, bad2 = timed_region()
        cfg["hint"] = False
        hinted_ok = None
        if bad2 == 0 and not args.no_parity:
            sync()
            od, oi = expected_by_oracle(last_query[0])
            hinted_ok = same_result(outs[0][0].cpu().numpy(), outs[0][1].cpu().numpy(), od, oi, tie_free_order=True)
            if not hinted_ok:
                raise SystemExit("PARITY FAILURE of the hinted one-stream steps against the oracle")
        step_flags[0], cfg["n_streams"] = keep
        single_stream["with_admission_hint"] = {
            "ms_per_step": round(1e3 * el2 / args.steps, 5), "achieved_GBps": round(alg_bytes_const / (1e-3 * 1e3 * el2 / args.steps) / 1e9, 1),
            "frac": round(alg_bytes_const / (el2 / args.steps) / 1e9 / HBM_PEAK_GBPS, 4), "status_max": bad2, "parity_vs_oracle": hinted_ok,
            "label": "UPPER BOUND of what a hint can buy: an oracle-grade hint (the query's own exact k-th acc, known from an untimed call) -- "
                     "no caller has it; the hint a caller CAN have (the previous query date's d_k, PathShadowing(hint=\"auto\")) is "
                     "timed at the API in blocking_shadow_api.rolling_dates, with held / fell-short counts",
            "hint": f"psh_profile.tau_hint = every query's exact k-th acc x {HINT_MARGIN} (d_k known to {50 * (HINT_MARGIN - 1):.1f} %; "
                    "~2.6 k windows below it at any W)",
            "launches": ("psh::scan_fused_kernel<..,HINTED>: no sample phase, no first grid barrier" if W <= 33 else
                         "the three launches, the sample launch as one block that derives the level's constants")}

    windows_per_step = world * R * Tp * B
    value = windows_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel: algorithmic bytes / live-measured duration
    mx = (W <= (33 if B == 1 else 25) and args.filter != "valu")
    wt = "20" if W == 20 else "0"
    # the batched scan's rejection test as the library chooses it (psh_capi.hip: the 8-bit product from 32 queries on, unless
    # PSH_FLAG_MQ_F16; below that the f16 one) -- kernel name, MFMA count and peak below follow THIS, not the flag alone
    f16_batch = args.mq_f16 or B < 32
    # which path serves the call (the launch plan's answer) -- and, for the sharded run, ONE bracketed launch of the
    # local scan outside the timed loop (per-GPU kernel time; the timed loop itself carries no events there)
    info = {}
    one_ms = float("nan")
    if on_gpu:
        one = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for e in one:
            e.record()
        torch.cuda.synchronize()
        _native.scan_topk(ds[:, 0, :], q, k, h=h, workspace=ws, flags=flags, info=info, scan_events=one)
        torch.cuda.synchronize()
        one_ms = one[0].elapsed_time(one[1])
    fused = info.get("path") == 2
    overlap_mode = info.get("path") == 3
    kernel_name = ("psh::stream_scan_long_kernel<true,%d> (the scan of the three overlap-friendly launches for a LONG window, up to three "
                   "queries a pass: the f16 banded product as a K-loop over %d steps of 16 taps, 1 + queries MFMAs each, + exact fp32 "
                   "recheck from memory)" % (min(B, 3), (W + 31 + 15) // 16)
                   if (overlap_mode and W > 33) else
                   "psh::stream_scan_kernel<%s,true> (the scan of the three overlap-friendly launches: f16 matrix-core rejection "
                   "test + exact fp32 recheck over the whole ensemble, no barrier; psh::stream_sample_kernel before it and "
                   "psh::stream_rank_kernel behind it run beside the scans of the other streams)" % wt if overlap_mode else
                   "psh::scan_fused_kernel<%s,true> (the WHOLE step in one launch: bootstrap, threshold, f16 matrix-core "
                   "rejection test + exact fp32 recheck over the ensemble, distributed selection)" % wt if fused else
                   ("psh::scan_mx_kernel<%s,true>" if B == 1 else "psh::scan_mq_kernel<%s,true>" if f16_batch else "psh::scan_mq8_kernel<%s,true>") % wt
                   + (" (full scan: f16 matrix-core rejection test + exact fp32 recheck)" if (B == 1 or f16_batch) else
                      " (full scan: 8-bit matrix-core rejection test, one v_mfma_i32_32x32x32_i8 per tile, + exact fp32 recheck)") if mx
                   else "psh::scan_kernel<%s,true,1> (full scan, VALU rejection test)" % wt)
    alg_bytes = R * T * 4 + B * W * 4 + B * k * 12          # one read of the ensemble + query + result
    roofline = None
    if sharded is None:
        avg_ms = float(np.mean(scan_ms))
        # Launches of different streams are CO-RESIDENT in overlap mode (the blocks of scan i+1 take over a compute unit
        # when scan i's block leaves it): a launch's begin-to-end duration then spans two scans sharing the chip, and the
        # sum of the durations exceeds the wall time.  What a launch costs is the interval between the ENDS of consecutive
        # scan launches, from the same HIP events; with one stream the two figures coincide up to the launch gap.
        interval_ms = float(np.mean(end_gaps_ms)) if (overlap_mode and end_gaps_ms) else None
        achieved = alg_bytes / ((interval_ms if interval_ms else avg_ms) * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, the guide's gfx950 correction): they
        # cannot be collected inside this run (PMC passes are their own rocprofv3 runs, tools/profile_round.sh), so the
        # figure is READ from the committed summary of those passes and labelled as such
        traffic = None
        traffic_source = None
        tfile = REPO / "profiles" / "hbm_traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                if (tj.get("workload") == f"R={R},T={T},W={W},h={h},k={k},B={B}" and ("scan_fused" in tj.get("kernel", "")) == fused
                        and ("stream_scan" in tj.get("kernel", "")) == overlap_mode):
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)"
            except Exception:   # noqa: BLE001
                traffic = None
        # `achieved` / `frac`: algorithmic bytes over the WHOLE timed region's time per step (every launch of it, every gap --
        # what the driver's clock sees); the figure from the bracketed launches (every EV_EVERY-th) rides beside it
        whole = alg_bytes / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(whole, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(whole / HBM_PEAK_GBPS, 4),
                    "achieved_bracketed_launches": round(achieved, 1), "frac_bracketed_launches": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": alg_bytes,
                    "avg_launch_ms": round(avg_ms, 5), "min_launch_ms": round(float(np.min(scan_ms)), 5),
                    "launches_timed": len(scan_ms), "launches_timed_every": EV_EVERY,
                    "frac_of_measured_copy_rate": round(whole / HBM_COPY_GBPS, 4)}
        if B > 1 and mx and not overlap_mode:
            # A batch is NOT bandwidth-bound: every ensemble byte is used by B queries (physical HBM traffic is one read of
            # the ensemble per launch: 0.5 GB in ~4 ms).  What bounds scan_mq_kernel is the matrix cores plus the vector ALUs
            # behind them ("report VALU utilisation + effective GB/s"): per 1024-window segment and group of four
            # queries 8 v_mfma_f32_32x32x16_f16 (4 tiles x K = 32: 20 taps + 7 shifts of the 8-window rows) and a 34-v_min3
            # epilogue.  `achieved` = ISSUED matrix-core flop/s (2 x 32 x 32 x 16 per MFMA; 62.5 % of the MACs are useful
            # ones, 20 taps x 1024 (window, query) pairs per tile pair) against the guide's dense f16 peak; busy fractions of
            # the matrix cores and the vector ALUs come from the committed PMC pass of this command, when there is one.
            # Round 4: the test is an 8-BIT product by default (scan_mq8_kernel): ONE v_mfma_i32_32x32x32_i8 per tile (K = 32
            # takes the whole band), 4 per segment and group, priced against the guide's dense 8-bit rate (2x the f16 one);
            # --mq-f16 runs and prices the f16 kernel as before.
            nseg_b = (Tp + 1023) // 1024
            groups = (B + 3) // 4
            per_group = 8 if f16_batch else 4
            mfma_per_launch = R * nseg_b * (groups * per_group + 8)         # + the 8 (f16) window-energy MFMAs of a segment
            flops = R * nseg_b * (groups * per_group * 2 * 32 * 32 * (16 if f16_batch else 32) + 8 * 2 * 32 * 32 * 16)
            ach_tf = flops / (avg_ms * 1e-3) / 1e12
            MFMA_F16_DENSE_TF = 2500.0 if f16_batch else 5000.0            # MI355X_MICROARCH.md: ~2.5 PF dense bf16 / f16, 8-bit at twice that
            pmc = None
            pfile = REPO / "profiles" / "q512_pmc.json"
            if pfile.exists():
                try:
                    pj = json.loads(pfile.read_text())
                    if (pj.get("workload") == f"R={R},T={T},W={W},h={h},k={k},B={B}"
                            and ("scan_mq8_kernel" in pj.get("kernel", "")) == (not f16_batch)):       # (the counters of THIS kernel only)
                        pmc = pj
                except Exception:   # noqa: BLE001
                    pmc = None
            roofline = {"bound": "mfma+valu", "kernel": kernel_name, "achieved": round(ach_tf, 1), "peak": MFMA_F16_DENSE_TF,
                        "unit": "TFLOP/s" if f16_batch else "TOP/s (8-bit)", "frac": round(ach_tf / MFMA_F16_DENSE_TF, 4),
                        "mfma_per_launch": mfma_per_launch, "issued_flops_per_launch": flops, "useful_mac_fraction": 0.625,
                        "matrix_core_busy_frac": pmc.get("matrix_core_busy_frac") if pmc else None,
                        "valu_busy_frac": pmc.get("valu_busy_frac") if pmc else None,
                        "pmc_source": "profiles/q512_pmc.json (rocprofv3 --pmc passes of this command: SQ_VALU_MFMA_BUSY_CYCLES, "
                                      "SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES; not measured in this run)" if pmc else None,
                        "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
                        "effective_GBps_per_query_equivalent": round(B * R * T * 4 / (avg_ms * 1e-3) / 1e9, 1),
                        "physical_algorithmic_GBps": round(alg_bytes / (avg_ms * 1e-3) / 1e9, 1),
                        "avg_launch_ms": round(avg_ms, 5), "min_launch_ms": round(float(np.min(scan_ms)), 5),
                        "launches_timed": len(scan_ms), "launches_timed_every": EV_EVERY}
        if interval_ms:
            roofline["avg_launch_interval_ms"] = round(interval_ms, 5)
            roofline["note"] = ("`achieved` / `frac` = algorithmic bytes / ms_per_step of the whole timed region; overlap mode: "
                                "`achieved_bracketed_launches` = algorithmic bytes / avg_launch_interval_ms (end-to-end interval of consecutive "
                                "scan launches, HIP events on the launches' own streams); avg_launch_ms is a launch's begin-to-end "
                                "duration WHILE it shares the chip with the neighbouring scan (profiles/: the kernel trace gives both)")
    elif on_gpu:
        achieved = alg_bytes / (one_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(one_ms, 5),
                    "launches_timed": 1, "note": "per-GPU, rank 0, one bracketed launch of the local scan outside the timed loop"}

    stages = None
    cpu = None
    if rank == 0 and world == 1 and on_gpu:
        # per-stage HIP-event timings of the SEPARATE launches (the fused launch has no stages to bracket); one untimed call
        # first: kernels the timed steps never ran are loaded on their first launch
        _native.scan_topk(ds[:, 0, :], q, k, h=h, workspace=ws, flags=flags | _native.FLAG_NO_FUSE)
        torch.cuda.synchronize()
        _, _, _, stages = _native.scan_topk(ds[:, 0, :], q, k, h=h, workspace=ws, profile=True, flags=flags | _native.FLAG_NO_FUSE)
        stages = {key: (round(val, 5) if isinstance(val, float) else val) for key, val in stages.items()}
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(ds_host, q_host, k, h)
            cpu["value"] = round(cpu["value"], 1)

    blocking = None
    if rank == 0 and world == 1 and on_gpu and B == 1 and sharded is None and not args.no_blocking_api:
        blocking = blocking_shadow_api(sa, syn, ds, ds_host, q_hosts, W, h, k, None if args.no_parity else expected_by_oracle)

    if rank == 0:
        out = {
            "metric": f"windows scanned/sec (k-nearest-path scan, Identity + RelativeMSE, W={W}, k={k})",
            "value": round(value, 1), "unit": "windows/s", "n_gpus": world,
            "rccl_world_size": dist.get_world_size() if (use_pg and on_gpu) else None, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_label(R, T, W, h, k, B, world),
                       "R_per_gpu": R, "R_total": world * R, "T": T, "W": W, "horizon": h, "k": k, "queries": B,
                       "windows_per_step": windows_per_step,
                       "sharding": "rows (R) across ranks, local top-k + one all-gather + merge; consecutive steps "
                                   "(independent queries) pipelined: the all-gather of step i overlaps the scan of step i+1"
                                   if use_pg else "none",
                       "inputs_resident_in_hbm": True,
                       "issue": (f"consecutive steps are independent queries issued round-robin on {n_streams} HIP streams "
                                 "(PSH_FLAG_OVERLAP: sample + level, barrier-free scan, ranking per step)") if n_streams > 1 else "one stream"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "achieved_hbm_GBps_whole_step": round(world * alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "streams": n_streams,
            "ms_per_step_repeats": {"n": len(region_ms), "median": round(float(np.median(region_ms)), 5), "min": round(min(region_ms), 5),
                                    "max": round(max(region_ms), 5), "what": f"the timed region (warm-up {args.warmup} + {args.steps} steps) run "
                                    f"{len(region_ms)} times; `value` / `ms_per_step` are the first region's"},
            "single_stream_fused": single_stream,
            "blocking_shadow_api": blocking,
            "host_enqueue_ms_per_step": round(1e3 * host_enqueue / args.steps, 5),
            "stages_ms_separate_launches": stages,
            "parity_vs_reference_golden": parity,
            "parity_vs_golden": parity if parity is not None else parity_first_oracle,
            "parity_golden": golden_name if parity is not None else ("distributed oracle (no committed fixture for these sizes)"
                                                                    if parity_first_oracle is not None else None),
            "parity_vs_oracle_query_subset": parity_oracle,
            "parity_rotating_queries": rotating,
            "cpu_oracle_test_hook": (not on_gpu) or None,
            "fused_launch_gave_up_rerun_as_separate_launches": fused_retry,
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
