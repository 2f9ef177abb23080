"""The Foveal 'testing' workload of tools/bench_foveal.py alone (for rocprofv3 passes): python tools/fov_prof.py [flags] [R] [B] [k]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from shadowing_amd import _native, synthetic as syn  # noqa: E402
from shadowing_amd.path_embedding import Foveal  # noqa: E402

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
R = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
dev = torch.device("cuda", 0)
emb = Foveal(alpha=1.15, beta=0.9, max_context=126)
ker = emb.kernel[:, 0, :].contiguous().to(dev)
g = torch.Generator(device=dev).manual_seed(1)
ds = torch.randn((R, 4096), generator=g, device=dev) * 0.0126
x = torch.tensor(syn.gbm_log_returns((B, 126), 2))
hx = emb(x[:, None, :])[:, 0, :].contiguous().to(dev)
ws = _native.Workspace(dev)
for _ in range(4):
    out = _native.scan_topk_embedded(ds, ker, hx, k, h=252, workspace=ws, flags=flags, keep_plan=True)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    _native.scan_topk_embedded(ds, ker, hx, k, h=252, workspace=ws, flags=flags, keep_plan=True)
torch.cuda.synchronize()
print("ms per call", (time.perf_counter() - t0) * 100)
