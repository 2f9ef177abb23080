import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from shadowing_amd import _native, synthetic as syn
import oracle
oracle.build()
dev = torch.device("cuda", 0)
def run(R, T, d, K, B, k, h, seed, wav=False):
    ds = syn.dataset(R, T, seed)
    if wav:
        ker = syn.wavelet_bank((d - 1) // 2, K)
    else:
        ker = (np.random.default_rng(seed).standard_normal((d, K)) * 0.3).astype(np.float32)
    d = ker.shape[0]
    x = syn.gbm_log_returns((B, K), seed + 1)
    hx = torch.nn.functional.conv1d(torch.tensor(x)[:, None, :], torch.tensor(ker)[:, None, :])[:, :, 0].contiguous()
    dsd, kd, hd = torch.tensor(ds[:, 0, :]).to(dev), torch.tensor(ker).to(dev), hx.to(dev)
    outs = {}
    for name, fl in (("mx", _native.FLAG_EMBED_MX), ("valu", 0)):
        info = {}
        dd, ii, st = _native.scan_topk_embedded(dsd, kd, hd, k, h=h, flags=fl)
        torch.cuda.synchronize()
        outs[name] = (dd.cpu().numpy(), ii.cpu().numpy(), st.cpu().numpy())
    same = np.array_equal(outs["mx"][0].view(np.uint32), outs["valu"][0].view(np.uint32)) and np.array_equal(outs["mx"][1], outs["valu"][1])
    od, oi = oracle.scan_topk_embedded(ds, ker, hx.numpy(), k, h=h)
    ok = {n: bool(np.array_equal(o[0].view(np.uint32), od.view(np.uint32)) and np.array_equal(o[1], oi)) for n, o in outs.items()}
    print(f"R={R} T={T} d={d} K={K} B={B} k={k} h={h}: mx==valu {same}; vs oracle {ok}; status mx {outs['mx'][2].max()} valu {outs['valu'][2].max()}", flush=True)
    if not ok["mx"]:
        a, b = outs["mx"], (od, oi)
        bad = np.nonzero(a[0].view(np.uint32) != b[0].view(np.uint32))
        print("   first diffs", bad[0][:5], bad[1][:5], a[0][bad][:5], b[0][bad][:5])
run(2048, 2048, 11, 252, 3, 200, 20, 1, wav=True)
run(4096, 1024, 5, 23, 2, 100, 7, 2)
run(3000, 1100, 12, 64, 5, 300, 3, 3)
run(5000, 515, 1, 7, 1, 50, 0, 4)
run(4096, 4096, 11, 252, 16, 1024, 20, 5, wav=True)
# timing: configs[4]
R, T, K, B, k, h = 32768, 4096, 252, 16, 1024, 20
g = torch.Generator(device=dev).manual_seed(1)
ds = torch.randn((R, T), generator=g, device=dev) * 0.0126
wk = torch.tensor(syn.wavelet_bank(5, 252))
xq = torch.tensor(syn.rolling_queries(B, 252, 2))
hxw = torch.nn.functional.conv1d(xq[:, None, :], wk[:, None, :])[:, :, 0].contiguous().to(dev)
kw = wk.contiguous().to(dev)
ws = _native.Workspace(dev)
for name, fl in (("mx", _native.FLAG_EMBED_MX), ("valu", 0)):
    for Bq in (16, 1):
        out = _native.scan_topk_embedded(ds, kw, hxw[:Bq].contiguous(), k, h=h, workspace=ws, flags=fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = _native.scan_topk_embedded(ds, kw, hxw[:Bq].contiguous(), k, h=h, workspace=ws, flags=fl)
        torch.cuda.synchronize()
        print(name, "B", Bq, "ms", (time.perf_counter() - t0) / 3 * 1e3, "status", int(out[2].max()), flush=True)
