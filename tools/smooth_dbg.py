"""Debug probe: the smooth-ensemble cases of tests/test_gpu_parity.py through the three launches: status, candidates, the level."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from shadowing_amd import _native
dev = torch.device("cuda:0")
R, T, h, k = 8192, 2048, 0, 1024
for W in (40, 64, 126, 250):
    rng = np.random.default_rng(2100 + W)
    ds = (0.05 * np.cumsum(rng.standard_normal((R, T)), axis=1)).astype(np.float32)
    q = (0.05 * np.cumsum(rng.standard_normal((1, W)))).astype(np.float32).reshape(1, W)
    ds_t, q_t = torch.as_tensor(ds).to(dev), torch.as_tensor(q).to(dev)
    for skip in ("8", "0"):
        os.environ["PSH_STREAM_SKIP"] = skip
        info = {}
        ws = _native.Workspace(dev)
        d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws, info=info)
        torch.cuda.synchronize()
        lay = _native.candidates_layout(R, T, 1, W, h, k, ws.buf.numel())
        o = lay["hdr_stream_ncand"]
        tot = ws.buf[o:o + 4].view(torch.int32) if o >= 0 else None
        cap = lay["stream_cap"]
        nx = float((q.astype(np.float64) ** 2).sum())
        print("W", W, "skip", skip, "status", st.tolist(), info, "ncand", None if tot is None else int(tot[0]), "cap", cap, "d_k", float(d[0, -1]), "nx", nx, flush=True)
