"""Probe: where a wave of the wavelet scan (embed_mx_kernel, configs[4]: d = 11, K = 252, 16 queries) spends its shader cycles --
segment set-up and f16 conversion / banded product on the matrix cores / accumulators -> energies and A fragments / per-query pass
(s_memtime stamps of the instrumented build, per wave, summed over its half segments)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from shadowing_amd import _build
if "PSH_LIB" not in os.environ:
    os.environ["PSH_LIB"] = str(_build.build(tuning=True))
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
R, T, B, k, h = 32768, 4096, 16, 1024, 20
g = torch.Generator(device=dev).manual_seed(1)
ds = torch.randn((R, T), generator=g, device=dev) * 0.0126
wk = torch.tensor(syn.wavelet_bank(5, 252))
xq = torch.tensor(syn.rolling_queries(B, 252, 2))
hxw = torch.nn.functional.conv1d(xq[:, None, :], wk[:, None, :])[:, :, 0].contiguous().to(dev)
kw = wk.contiguous().to(dev)
ws = _native.Workspace(dev)
buf = torch.zeros(256 * 8 * 6, dtype=torch.int64, device=dev)
os.environ["PSH_DBG_TIMES_PTR"] = str(buf.data_ptr())
for _ in range(3):
    out = _native.scan_topk_embedded(ds, kw, hxw, k, h=h, workspace=ws, flags=_native.FLAG_EMBED_MX, profile=True)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(-1, 6).astype(np.float64)
t, cnt = raw[:, :4], raw[:, 4:]
print("scan_ms", round(out[3]["scan_ms"], 4), "grid", out[3]["grid_blocks"])
tot = t.sum(1)
print("cycles per wave (mean over %d waves): set-up %.0f  product %.0f  energies + per-query pass %.0f  verification %.0f  total %.0f" % ((len(t),) + tuple(t.mean(0)) + (tot.mean(),)))
n_half = 2 * R * 4 / len(t)
print("per half segment: set-up %.0f  product %.0f  energies + per-query pass %.0f  verification %.0f   (product floor: 216 MFMAs x 16 = 3456)" % tuple(t.mean(0) / n_half))
print("survivors verified per segment %.2f, passes of <= 4 per segment %.2f" % tuple(cnt.sum(0) / (R * 4)))
