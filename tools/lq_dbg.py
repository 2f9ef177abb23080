import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn
import oracle
dev=torch.device('cuda:0')
for (W,B,h,k) in [(126, 17, 0, 64)]:
    R,T=4096,2048
    ds = syn.dataset(R, T, 2200 + W)
    q = syn.gbm_log_returns((B, W), 2300 + W + B)
    ds_t = torch.as_tensor(np.ascontiguousarray(ds[:, 0, :])).to(dev); q_t=torch.as_tensor(q).to(dev)
    od, oidx = oracle.scan_topk(ds, q, k, h=h)
    xn2 = (q.astype(np.float64) ** 2).sum(axis=1)
    lev = ((od[:, k - 1].astype(np.float64) ** 2) * xn2 * 1.1).astype(np.float32)
    ws = _native.Workspace(dev)
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, tau_hint=torch.as_tensor(lev).to(dev), workspace=ws)
    torch.cuda.synchronize()
    print("hinted status", st.tolist())
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws)
    torch.cuda.synchronize()
    print("sampled status", st.tolist())
    # the levels the threshold kernel chose vs the k-th acc
    lay = _native.candidates_layout(R, T, B, W, h, k, ws.buf.numel())
    qs = ws.buf[lay["qstate"]:lay["qstate"] + 48 * B].view(torch.float32).view(B, 12).cpu().numpy()
    tau = qs[:, 1].view(np.uint32).view(np.float32)
    kth = (od[:, k - 1].astype(np.float64) ** 2) * xn2
    print("tau / k-th acc:", np.round(tau / kth, 3))
    d, idx, st = _native.scan_topk(ds_t, q_t, k, h=h, workspace=ws, flags=_native.FLAG_FILTER_VALU)
    torch.cuda.synchronize()
    qs = ws.buf[lay["qstate"]:lay["qstate"] + 48 * B].view(torch.float32).view(B, 12).cpu().numpy()
    tau_v = qs[:, 1].view(np.uint32).view(np.float32)
    print("VALU pipeline: status", st.tolist(), "tau / k-th acc:", np.round(tau_v / kth, 3))
    print("lq tau / valu tau:", np.round(tau / tau_v, 3))
