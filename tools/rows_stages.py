"""Stage times of the scan of one-window rows (RelativeMSE.forward_topk's N pre-embedded points), warm calls."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shadowing_amd import _native
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for (N, d, B, k) in [(4194304, 34, 1, 1024), (4194304, 34, 6, 8192), (4194304, 20, 1, 1024)]:
    y = torch.randn((N, d), generator=g, device=dev) * 0.02
    x = torch.randn((B, d), generator=g, device=dev) * 0.02
    ws = _native.Workspace(dev)
    for _ in range(4): out = _native.scan_topk(y, x, k, h=0, workspace=ws, profile=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): _native.scan_topk(y, x, k, h=0, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    print(N, d, B, k, {a: round(b, 4) if isinstance(b, float) else b for a, b in out[3].items()}, "stream ms/call", round(e0.elapsed_time(e1) / 50, 4), "status", int(out[2].max()))
