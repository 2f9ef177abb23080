"""What one stream's step cannot go below: the fused single launch with its scan loop REMOVED (tuning build, PSH_DBG=16:
every wave scans the ONE unit it requested before the admission level was known -- 4096 of the 131072 units -- and the
launch keeps everything else: launch ramp, sample, first barrier, level, second barrier, candidate exchange, ranking),
timed back to back on one stream beside the full launch.  step >= skeleton + (ensemble bytes - skeleton's bytes) / streaming rate.

    python -m shadowing_amd._build --tuning && PSH_LIB=shadowing_amd/lib/libpsh_hip_tuning.so python tools/fused_skeleton.py
"""
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("PSH_LIB", str(REPO / "shadowing_amd/lib/libpsh_hip_tuning.so"))
import torch  # noqa: E402
from shadowing_amd import _native, synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
R, T, W, h, k = 32768, 4096, 20, 20, 1024
ds = torch.as_tensor(syn.dataset(R, T, 0)[:, 0, :].copy()).to(dev)
q = torch.as_tensor(syn.single_query(W, 1)[None, :].copy()).to(dev)
ws = _native.Workspace(dev)
out = (torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1, k, 2), dtype=torch.int32, device=dev),
       torch.zeros((1,), dtype=torch.int32, device=dev))


def run(n):
    for _ in range(20):
        _native.scan_topk(ds, q, k, h=h, workspace=ws, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        _native.scan_topk(ds, q, k, h=h, workspace=ws, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


res = {}
for rep in range(3):
    os.environ["PSH_DBG"] = "0"
    res.setdefault("full_us", []).append(round(run(1000), 2))
    os.environ["PSH_DBG"] = "16"
    res.setdefault("skeleton_us", []).append(round(run(1000), 2))
    ws.arm()                                               # (the skeleton's RETRY disarms nothing, but start clean)
os.environ["PSH_DBG"] = "0"
full, skel = min(res["full_us"]), min(res["skeleton_us"])
alg = R * T * 4
skel_bytes = 4096 * (1024 + W - 1) * 4 * 2                 # the sampled units + one scanned unit per wave
for rate in (6.7e12, 8.0e12):
    res[f"floor_us_at_{rate / 1e12:.1f}TBps"] = round(skel + (alg - skel_bytes) / rate * 1e6, 2)
res.update(full_best_us=full, skeleton_best_us=skel, scan_loop_share_us=round(full - skel, 2),
           note="one stream, back-to-back launches; skeleton = PSH_DBG=16 (one unit per wave instead of the ensemble)")
print(json.dumps(res))
