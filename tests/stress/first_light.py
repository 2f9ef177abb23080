"""Development aid: run the HIP path on a few cases, report mismatches vs the oracle and
stage timings.  (Not part of the product or the test suite.)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import oracle
from shadowing_amd import _native, synthetic as syn

dev = torch.device("cuda", 0)
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")

def run(R, T, W, h, k, B, seed=0, exhaustive=False, check=True, reps=0):
    ds = syn.dataset(R, T, seed); q = syn.gbm_log_returns((B, W), seed + 1)
    ds_t = torch.as_tensor(ds[:, 0, :].copy()).to(dev); q_t = torch.as_tensor(q).to(dev)
    ws = _native.Workspace(dev)
    d, idx, st, prof = _native.scan_topk(ds_t, q_t, k, h=h, exhaustive=exhaustive, profile=True, workspace=ws)
    torch.cuda.synchronize()
    d, idx, st = d.cpu().numpy(), idx.cpu().numpy(), st.cpu().numpy()
    msg = f"R={R} T={T} W={W} h={h} k={k} B={B} exh={exhaustive}: status={st.tolist()[:4]} prof={ {a: (round(b,4) if isinstance(b,float) else b) for a,b in prof.items()} }"
    if check:
        od, oidx = oracle.scan_topk(ds, q, k, h=h)
        okd = np.array_equal(d.view(np.uint32), od.view(np.uint32)); oki = np.array_equal(idx, oidx)
        msg += f" | d_exact={okd} idx_exact={oki}"
        if not (okd and oki):
            bad = np.argwhere(d.view(np.uint32) != od.view(np.uint32))
            msg += f" first_bad={bad[:3].tolist()} got={d[0,:4]} {idx[0,:4].tolist()} want={od[0,:4]} {oidx[0,:4].tolist()}"
    print(msg, flush=True)
    if reps:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            _native.scan_topk(ds_t, q_t, k, h=h, exhaustive=exhaustive, workspace=ws)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        N = R * (T - W - h + 1) * B
        print(f"   {dt*1e6:.1f} us/step  {N/dt:.3e} windows/s  alg {R*T*4/dt/1e9:.1f} GB/s", flush=True)

run(64, 1024, 20, 20, 64, 1)
run(64, 1024, 20, 20, 64, 1, exhaustive=True)
run(300, 1100, 20, 20, 128, 3)
run(50, 515, 20, 7, 33, 2)
run(40, 600, 37, 5, 50, 2)
run(40, 600, 8, 3, 50, 2)
run(4096, 4096, 20, 20, 1024, 1, seed=7, reps=20)
run(32768, 4096, 20, 20, 1024, 1, seed=0, check=True, reps=50)
run(32768, 4096, 20, 20, 1024, 8, seed=0, check=False, reps=5)
