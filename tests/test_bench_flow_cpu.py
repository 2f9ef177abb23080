"""bench.py's OWN row-sharded control flow on CPU: two ranks over gloo through `torch.distributed.run`, exactly the
command line the driver uses for N > 1, with the `--cpu-oracle` test hook (the CPU oracle stands in for the HIP scan;
nothing is measured).  A Python-level bug in step / drain / timed_region / the parity checks / the JSON line shows up
here instead of on a multi-GPU lease.  Plus the sweep report and the host-side checker helpers on canned data."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(nproc: int, extra: list[str], timeout: int = 600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(REPO / "bench.py"), "--gpus", str(nproc), "--cpu-oracle"] + extra
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "2"
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(REPO))
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, lines


def test_two_rank_gloo_run_of_bench_flow(oracle_mod):
    res, lines = _run_bench(2, ["--steps", "7", "--warmup", "3", "--rows-per-gpu", "24", "--T", "160", "--k", "50"])
    assert res.returncode == 0, res.stderr[-3000:]
    assert len(lines) == 1, res.stdout                      # rank 0 prints ONE JSON line, the other rank none
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 7 and j["warmup"] == 3 and j["scaling"] == "weak"
    assert j["cpu_oracle_test_hook"] is True and j["rccl_world_size"] is None
    assert j["config"]["R_total"] == 48 and j["config"]["windows_per_step"] == 2 * 24 * (160 - 20 - 20 + 1)
    # the first merged result against the distributed oracle, and what the timed steps left behind against it
    assert j["parity_vs_golden"] is True
    assert j["parity_rotating_queries"]["ok"] is True and len(j["parity_rotating_queries"]["query_batches_checked"]) == 2
    assert j["value"] > 0 and j["ms_per_step"] > 0


def test_two_rank_gloo_run_with_a_batch_and_a_short_shard(oracle_mod):
    # a batch of queries, and shards with fewer windows than k (2 rows x 9 windows < 40): the padded general path
    res, lines = _run_bench(2, ["--steps", "3", "--warmup", "1", "--rows-per-gpu", "2", "--T", "48", "--k", "30", "--queries", "3"])
    assert res.returncode == 0, res.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["parity_rotating_queries"]["ok"] is True


def test_four_rank_gloo_run_with_odd_shards(oracle_mod):
    # four ranks (the driver's N = 4 line), an odd number of rows per rank and a row length that is no multiple of anything,
    # k above a shard's window count: every rank pads, four lists meet in the merge
    res, lines = _run_bench(4, ["--steps", "5", "--warmup", "2", "--rows-per-gpu", "7", "--T", "131", "--k", "600"])
    assert res.returncode == 0, res.stderr[-3000:]
    assert len(lines) == 1, res.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 4 and j["config"]["R_total"] == 28 and j["scaling"] == "weak"
    assert j["parity_vs_golden"] is True and j["parity_rotating_queries"]["ok"] is True
    assert "4 ways" in j["config"]["workload"] or "not a BASELINE" in j["config"]["workload"]


def test_two_rank_gloo_run_with_512_query_dates(oracle_mod):
    # configs[2]'s batch size through the sharded flow (B*k*12 bytes per rank in the one all-gather: 6 MiB at k = 1024; here small)
    res, lines = _run_bench(2, ["--steps", "2", "--warmup", "1", "--rows-per-gpu", "6", "--T", "120", "--k", "24", "--queries", "512"])
    assert res.returncode == 0, res.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["config"]["queries"] == 512 and j["parity_rotating_queries"]["ok"] is True
    assert j["config"]["windows_per_step"] == 2 * 6 * (120 - 20 - 20 + 1) * 512


def test_one_rank_hook_and_launcher_mismatch(oracle_mod):
    res, lines = _run_bench(1, ["--steps", "2", "--warmup", "1", "--rows-per-gpu", "8", "--T", "100", "--k", "16"])
    assert res.returncode == 0, res.stderr[-3000:]
    assert json.loads(lines[0])["parity_vs_golden"] is True
    # --gpus that disagrees with the launcher's world size is refused before anything runs
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(REPO / "bench.py"), "--gpus", "4", "--cpu-oracle", "--steps", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(REPO))
    assert res.returncode != 0 and "WORLD_SIZE=2" in (res.stderr + res.stdout)


def _line(n, value, ms, world=None, parity=True, rot=True):
    return {"n_gpus": n, "rccl_world_size": world if world is not None else (n if n > 1 else None), "ms_per_step": ms,
            "value": value, "unit": "windows/s", "parity_vs_golden": parity, "parity_rotating_queries": {"ok": rot}}


def test_sweep_report_on_canned_lines():
    import bench
    rows = bench.sweep_report([_line(1, 1.5e12, 0.0886), _line(2, 2.85e12, 0.0933), _line(4, 5.4e12, 0.0985),
                               _line(8, 9.6e12, 0.1108)])
    assert len(rows) == 4
    effs = [float(r.split("weak_scaling_efficiency=")[1].split()[0]) for r in rows]
    assert effs == [1.0, 0.95, 0.9, 0.8]
    assert all("PARITY" not in r and "MISMATCH" not in r and "unchecked" not in r for r in rows)
    assert "n_gpus=8 rccl_world_size=8 ms_per_step=0.1108" in rows[3]
    # lines that did not verify themselves are marked: their numbers are not evidence
    rows = bench.sweep_report([_line(1, 1e12, 0.1), _line(2, 2e12, 0.1, world=1), _line(4, 4e12, 0.1, parity=False),
                               _line(8, 8e12, 0.1, rot=False), _line(8, 8e12, 0.1, parity=None)])
    assert "RCCL-WORLD-MISMATCH" in rows[1] and "PARITY-FAILED" in rows[2] and "PARITY-FAILED" in rows[3]
    assert "parity-unchecked" in rows[4] and "PARITY-FAILED" not in rows[4]
    # efficiency is relative to the FIRST line, whatever its rank count
    rows = bench.sweep_report([_line(2, 2e12, 0.1), _line(8, 6e12, 0.13)])
    assert "weak_scaling_efficiency=0.750" in rows[1]


def test_host_merge_and_same_result():
    import bench
    g = np.random.default_rng(3)
    d = g.random((2, 40)).astype(np.float32)
    d[0, 5] = d[0, 17]                                       # an exact tie: (r, t) decides
    idx = np.stack([g.integers(0, 100, (2, 40)), g.integers(0, 50, (2, 40))], axis=-1).astype(np.int32)
    idx[1, 7, 0] = -1                                        # padding never wins
    d[1, 7] = 0.0
    md, mi = bench.host_merge(d, idx, 10)
    for b in range(2):
        real = idx[b, :, 0] >= 0
        order = sorted(range(40), key=lambda i: (d[b, i], idx[b, i, 0], idx[b, i, 1]))
        order = [i for i in order if real[i]][:10]
        assert np.array_equal(md[b], d[b, order]) and np.array_equal(mi[b], idx[b, order])
    assert bench.same_result(md, mi, md.copy(), mi.copy(), tie_free_order=True)
    # the reference's arbitrary order among exact ties: compared as sorted distances + the set of pairs
    perm = g.permutation(10)
    assert bench.same_result(md, mi, md[:, perm], mi[:, perm], tie_free_order=False)
    assert not bench.same_result(md, mi, md[:, perm], mi[:, perm], tie_free_order=True)
    wrong = mi.copy()
    wrong[0, 3, 1] += 1
    assert not bench.same_result(md, wrong, md, mi, tie_free_order=True)
    assert not bench.same_result(md, wrong, md[:, perm], mi[:, perm], tie_free_order=False)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", REPO / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workload_label_follows_the_sizes_and_the_rank_count():
    """config.workload names the BASELINE.json configuration the run IS: configs[1] only at its sizes on one GPU, configs[3]
    for the 8-way sharded run, configs[2] for 512 queries -- and says so when the sizes are not a BASELINE configuration
    (r04's line for --rows-per-gpu 131072 carried configs[1]'s name)."""
    wl = _bench_module().workload_label
    assert wl(32768, 4096, 20, 20, 1024, 1, 1).startswith("BASELINE.json configs[1]:")
    l8 = wl(32768, 4096, 20, 20, 1024, 1, 8)
    assert l8.startswith("BASELINE.json configs[3]:") and "R=262144" in l8 and "8 ways" in l8
    l2 = wl(32768, 4096, 20, 20, 1024, 1, 2)
    assert "configs[3]" in l2 and "2 of its 8 GPUs" in l2 and "R=65536" in l2 and not l2.startswith("BASELINE.json configs[1]")
    assert wl(32768, 4096, 20, 20, 1024, 512, 1).startswith("BASELINE.json configs[2]:")
    for other in (wl(131072, 4096, 20, 20, 1024, 1, 1), wl(32768, 4096, 64, 20, 1024, 1, 1), wl(24, 160, 20, 20, 50, 1, 2),
                  wl(32768, 4096, 20, 20, 1024, 40, 1)):
        assert other.startswith("not a BASELINE.json configuration") and "configs[1]" not in other
    assert "R=131072" in wl(131072, 4096, 20, 20, 1024, 1, 1)


@pytest.mark.parametrize("nproc", [1, 2])
def test_emitted_json_names_its_workload(oracle_mod, nproc):
    res, lines = _run_bench(nproc, ["--steps", "2", "--warmup", "1", "--rows-per-gpu", "24", "--T", "160", "--k", "50"])
    assert res.returncode == 0, res.stderr[-3000:]
    j = json.loads(lines[0])
    w = j["config"]["workload"]
    assert w.startswith("not a BASELINE.json configuration") and "R=24 " in w and f"{nproc} GPU" in w
    assert ("sharded 2 ways" in w) == (nproc == 2)
    assert "W=20" in j["metric"] and "k=50" in j["metric"]


def test_self_launch_refuses_more_ranks_than_gpus_with_one_line(monkeypatch, capsys):
    """`python bench.py --gpus 8` on a node that shows fewer GPUs: one clear line and exit code 2, no rank is started."""
    m = _bench_module()
    import subprocess as sp
    started = []
    monkeypatch.setattr(m, "_run_streaming", lambda *a, **k: started.append(a) or (_ for _ in ()).throw(AssertionError("a rank was launched")))
    monkeypatch.setattr(m.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(m.torch.cuda, "device_count", lambda: 1)
    assert m.self_launch(8, ["--gpus", "8"]) == 2
    err = capsys.readouterr().err
    assert err.count("\n") == 1 and "needs 8 visible GPUs" in err and "shows 1" in err and not started


def test_self_launch_retries_a_taken_rendezvous_port(monkeypatch, capsys):
    m = _bench_module()
    import subprocess as sp
    calls = []

    def fake_run(cmd, env):
        calls.append(cmd[cmd.index("--master-port") + 1])
        return (1, "RuntimeError: ... EADDRINUSE ...\n", 2.0) if len(calls) == 1 else (0, "", 5.0)
    monkeypatch.setattr(m, "_run_streaming", fake_run)
    assert m.self_launch(2, ["--gpus", "2", "--cpu-oracle"]) == 0
    assert len(calls) == 2
    assert "retrying" in capsys.readouterr().err
    # the same text from a run that died LATE (its ranks had their GPUs for minutes) is not a reason to run it again
    calls.clear()
    monkeypatch.setattr(m, "_run_streaming", lambda cmd, env: (calls.append(1) or (1, "... Address already in use ...\n", 300.0)))
    assert m.self_launch(2, ["--gpus", "2", "--cpu-oracle"]) == 1 and len(calls) == 1
