"""Probe: where a wave of the 8-bit batched scan (scan_mq8_kernel, configs[2]: 512 queries) spends its shader cycles -- the work
per segment before the group loop (staging, quantisation, energies, fragments, levels) / the group loop / the drain of the
survivor queue and the hand-over to the next unit (s_memtime stamps of the instrumented build, per wave)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from shadowing_amd import _build
if "PSH_LIB" not in os.environ:
    os.environ["PSH_LIB"] = str(_build.build(tuning=True))
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev)
q = torch.as_tensor(syn.rolling_queries(512, 20, 1)).to(dev)
ws = _native.Workspace(dev)
buf = torch.zeros(256 * 8 * 3, dtype=torch.int64, device=dev)
os.environ["PSH_DBG_TIMES_PTR"] = str(buf.data_ptr())
for _ in range(3):
    out = _native.scan_topk(ds, q, 1024, h=20, workspace=ws, profile=True)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 3).astype(np.float64)
t = t[t.sum(1) > 0]
prof = out[-1]
print("scan_ms", round(prof["scan_ms"], 4), "waves", len(t))
m = t.mean(0)
print("s_memtime ticks per wave (mean): set-up %.0f  group loop %.0f  drain + hand-over %.0f  total %.0f" % (m[0], m[1], m[2], m.sum()))
print("shares: set-up %.1f %%  group loop %.1f %%  drain + hand-over %.1f %%" % tuple(100 * m / m.sum()))
print("per segment (64 per wave) and per group (128 per segment): set-up %.0f, loop %.0f = %.1f per group, drain %.0f ticks" % (m[0] / 64, m[1] / 64, m[1] / 64 / 128, m[2] / 64))
tb = t.sum(1).reshape(-1, 8).max(1)                         # a block is done when its slowest wave is
print("blocks' times (ticks, the slowest wave of each): min %.0f  mean %.0f  max %.0f  -> the launch waits %.1f %% beyond the mean block" % (tb.min(), tb.mean(), tb.max(), 100 * (tb.max() / tb.mean() - 1)))
