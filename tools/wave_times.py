import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from shadowing_amd import _build
os.environ["PSH_LIB"] = str(_build.build(tuning=True))   # the instrumented build (-DPSH_TUNING): env overrides, time stamps
from shadowing_amd import _native, synthetic as syn
dev = torch.device("cuda", 0)
ds = torch.as_tensor(syn.dataset(32768, 4096, 0)[:, 0, :].copy()).to(dev); q = torch.as_tensor(syn.single_query(20, 1)[None]).to(dev)
ws = _native.Workspace(dev)
buf = torch.zeros(2 * 8192, dtype=torch.int64, device=dev)
for bpc in (1, 2):
    os.environ["PSH_BLOCKS_PER_CU"] = str(bpc)
    for rep in range(2):
        os.environ["PSH_DBG_TIMES_PTR"] = str(buf.data_ptr())
        *_, prof = _native.scan_topk(ds, q, 1024, h=20, workspace=ws, profile=True)
    t = buf.cpu().numpy().reshape(-1, 2)[: prof["grid_blocks"] * 16].astype(np.float64) * 10.0  # ns
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 1e3, (t[:, 1] - t0) / 1e3
    dur = en - st
    print(f"bpc={bpc} grid={prof['grid_blocks']} scan_ms={prof['scan_ms']:.4f}: start p0/p50/p100 = {np.percentile(st,[0,50,100]).round(1)} us; end p0/p10/p50/p90/p100 = {np.percentile(en,[0,10,50,90,100]).round(1)}; dur p0/p50/p100 = {np.percentile(dur,[0,50,100]).round(1)}")
    # by XCD (block b -> XCD b%8)
    blk = np.arange(len(en)) // 16
    for x in range(8):
        m = (blk % 8) == x
        print(f"   xcd{x}: end mean {en[m].mean():.1f} max {en[m].max():.1f}  dur mean {dur[m].mean():.1f}")
