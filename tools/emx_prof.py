import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
R, T, K, B, k, h = 32768, 4096, 252, int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 20
FL = _native.FLAG_EMBED_MX | (int(sys.argv[3]) if len(sys.argv) > 3 else 0)
g = torch.Generator(device=dev).manual_seed(1)
ds = torch.randn((R, T), generator=g, device=dev) * 0.0126
wk = torch.tensor(syn.wavelet_bank(5, 252))
xq = torch.tensor(syn.rolling_queries(B, 252, 2))
hxw = torch.nn.functional.conv1d(xq[:, None, :], wk[:, None, :])[:, :, 0].contiguous().to(dev)
kw = wk.contiguous().to(dev)
ws = _native.Workspace(dev)
for _ in range(3):
    out = _native.scan_topk_embedded(ds, kw, hxw, k, h=h, workspace=ws, flags=FL)
torch.cuda.synchronize()
out = _native.scan_topk_embedded(ds, kw, hxw, k, h=h, workspace=ws, flags=FL, profile=True)
print({k2: round(v, 4) if isinstance(v, float) else v for k2, v in out[3].items()})

import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    _native.scan_topk_embedded(ds, kw, hxw, k, h=h, workspace=ws, flags=FL)
torch.cuda.synchronize(); print("B", B, "k", k, "flags", FL, "ms per call", (time.perf_counter() - t0) * 100)
