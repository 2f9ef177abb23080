#!/usr/bin/env python
"""Timing of the embedded scan (psh_scan_topk_embedded) on the two Foveal workloads the
reference itself runs:
  tutorial : tutorial.ipynb cell 8   -- Foveal(1.15, 0.9, 126), horizon 252, k = 8192, B = 6, R = 2048, T = 4096
  testing  : testing.ipynb:95-104    -- the same embedding, horizon 252, k = 10000, B = 1, R = 131072, T = 4096
             (the reference prints 2.65 s per predict() call for it on an unnamed NVIDIA GPU, host copy included)
Prints one JSON line per workload: native ms per call with the ensemble resident in HBM,
the whole shadow() call through the reference API (host query -> host results), and the
generic torch formulation on the same device for scale.

    python tools/bench_foveal.py [--which tutorial testing] [--steps 20] [--rows 131072]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from shadowing_amd import _native, synthetic as syn  # noqa: E402
from shadowing_amd.path_embedding import Foveal, PathEmbedding, PredictionContext  # noqa: E402
from shadowing_amd.path_distance import RelativeMSE  # noqa: E402
from shadowing_amd.path_shadowing import PathShadowing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", nargs="+", default=["tutorial", "testing"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rows", type=int, default=0, help="override R of the testing workload")
    ap.add_argument("--k", type=int, default=0, help="override k (k = 8: hardly any survivor, the cheap pass alone)")
    ap.add_argument("--generic", action="store_true", help="also time the generic torch path on the device")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    emb = Foveal(alpha=1.15, beta=0.9, max_context=126)
    ker = emb.kernel[:, 0, :].contiguous().to(dev)
    cfgs = {"tutorial": dict(R=2048, T=4096, B=6, k=8192, h=252),
            "testing": dict(R=args.rows or 131072, T=4096, B=1, k=10000, h=252)}
    cfgs["wavelet"] = dict(R=args.rows or 32768, T=4096, B=16, k=1024, h=20)     # BASELINE.json configs[4], one GPU's shard
    for name in args.which:
        c = cfgs[name]
        if args.k:
            c["k"] = args.k
        if name == "wavelet":
            # "wavelet conv, W=252 replacing Identity, batched queries": the linear stage of a scattering embedding
            # (synthetic.wavelet_bank), 16 rolling query dates; no suffix structure, so the dense chains
            wk = torch.tensor(syn.wavelet_bank(5, 252))
            g = torch.Generator(device=dev).manual_seed(1)
            ds = torch.randn((c["R"], 1, c["T"]), generator=g, device=dev) * 0.0126
            xq = torch.tensor(syn.rolling_queries(c["B"], 252, 2))
            hxw = torch.nn.functional.conv1d(xq[:, None, :], wk[:, None, :])[:, :, 0].contiguous().to(dev)
            kw = wk.contiguous().to(dev)
            ws = _native.Workspace(dev)
            mxf = _native.FLAG_EMBED_MX            # a dense kernel: rejection test on the matrix cores (what PathShadowing passes)
            for _ in range(3):                     # the stage times of a WARM call (the first one loads the kernels)
                out = _native.scan_topk_embedded(ds[:, 0, :], kw, hxw, c["k"], h=c["h"], workspace=ws, profile=True, flags=mxf)
            assert int(out[2].max()) == 0, "overflow"
            nst = max(2, args.steps // 4)
            ms_by = {}
            for label, fl in (("matrix_cores", mxf), ("valu_dense_chains", 0)):
                for _ in range(2):
                    _native.scan_topk_embedded(ds[:, 0, :], kw, hxw, c["k"], h=c["h"], workspace=ws, flags=fl)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(nst):
                    _native.scan_topk_embedded(ds[:, 0, :], kw, hxw, c["k"], h=c["h"], workspace=ws, flags=fl)
                torch.cuda.synchronize()
                ms_by[label] = (time.perf_counter() - t0) / nst * 1e3
            ms = ms_by["matrix_cores"]
            windows = c["R"] * (c["T"] - 252 - c["h"] + 1)
            print(json.dumps(dict(workload="wavelet (BASELINE.json configs[4], per GPU)", config=c,
                                  embedding=f"wavelet_bank(5, 252): d={wk.shape[0]}, {int((wk != 0).sum())} non-zero taps",
                                  ms_per_call=round(ms, 3), ms_per_call_valu_dense_chains=round(ms_by["valu_dense_chains"], 3),
                                  windows=windows, query_windows_per_s=windows * c["B"] / (ms * 1e-3),
                                  stages_ms={k2: round(v, 4) for k2, v in out[3].items() if k2.endswith("_ms")},
                                  n_candidates=out[3]["n_candidates"])))
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        ds = torch.randn((c["R"], 1, c["T"]), generator=g, device=dev) * 0.0126      # testing.ipynb uses torch.randn
        x = torch.tensor(syn.gbm_log_returns((c["B"], 126), 2))
        hx = emb(x[:, None, :])[:, 0, :].contiguous().to(dev)
        ws = _native.Workspace(dev)
        dsv = ds[:, 0, :]
        # (keep_plan: repeated calls with the same kernel tensor skip the plan launch, as PathShadowing's do)
        for _ in range(3):                         # the stage times of a WARM call (the first one loads the kernels)
            out = _native.scan_topk_embedded(dsv, ker, hx, c["k"], h=c["h"], workspace=ws, profile=True)
        assert int(out[2].max()) == 0, "overflow"
        stages = out[3]
        for _ in range(3):
            _native.scan_topk_embedded(dsv, ker, hx, c["k"], h=c["h"], workspace=ws, keep_plan=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _native.scan_topk_embedded(dsv, ker, hx, c["k"], h=c["h"], workspace=ws, keep_plan=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        # the same scan walking the taps (what kernels with a gap in their support take): PSH_FLAG_EMBED_TAPS
        for _ in range(2):
            _native.scan_topk_embedded(ds[:, 0, :], ker, hx, c["k"], h=c["h"], workspace=ws, flags=_native.FLAG_EMBED_TAPS)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(2, args.steps // 2)):
            _native.scan_topk_embedded(ds[:, 0, :], ker, hx, c["k"], h=c["h"], workspace=ws, flags=_native.FLAG_EMBED_TAPS)
        torch.cuda.synchronize()
        ms_taps = (time.perf_counter() - t0) / max(2, args.steps // 2) * 1e3
        windows = c["R"] * (c["T"] - 126 - c["h"] + 1)
        taps = int((emb.kernel != 0).sum())
        # through the reference API, dataset already a device tensor (resident)
        obj = PathShadowing(emb, RelativeMSE(), ds, PredictionContext(horizon=c["h"]))
        obj.shadow(x.numpy(), k=c["k"], cuda=True)
        t0 = time.perf_counter()
        for _ in range(5):
            d, paths, idx = obj.shadow(x.numpy(), k=c["k"], cuda=True)
        api_ms = (time.perf_counter() - t0) / 5 * 1e3
        assert obj.last_path == "hip"
        # the reference's own timed call (testing.ipynb:95-104): predict() = scan + gather + statistic + weighted moments
        from shadowing_amd.statistics import realized_variance
        to_predict = lambda p: realized_variance(p, Ts=[2, 7, 252], vol=False)     # noqa: E731
        # (device_predict=True: the statistic is evaluated where the paths are -- an opt-in since r02, a plain lambda is
        #  handed numpy arrays like the reference's; the host figure is taken beside it below)
        obj.predict(x.numpy(), k=c["k"], to_predict=to_predict, eta=0.1, cuda=True, device_predict=True)
        t0 = time.perf_counter()
        for _ in range(5):
            obj.predict(x.numpy(), k=c["k"], to_predict=to_predict, eta=0.1, cuda=True, device_predict=True)
        predict_ms = (time.perf_counter() - t0) / 5 * 1e3
        host_tp = lambda p: realized_variance(np.asarray(p), Ts=[2, 7, 252], vol=False)   # noqa: E731
        t0 = time.perf_counter()
        obj.predict(x.numpy(), k=c["k"], to_predict=host_tp, eta=0.1, cuda=True)
        predict_host_ms = (time.perf_counter() - t0) * 1e3
        line = dict(workload=name, config=c, embedding="Foveal(1.15,0.9,126) d=34", ms_per_call=round(ms, 4),
                    ms_per_call_tap_walk=round(ms_taps, 4),
                    windows=windows, query_windows_per_s=windows * c["B"] / (ms * 1e-3),
                    nonzero_taps=taps, gfma_per_s=windows * taps * ((c["B"] + 2) // 3) / (ms * 1e-3) / 1e9,
                    stages_ms={k: round(v, 4) for k, v in stages.items() if k.endswith("_ms")},
                    n_candidates=stages["n_candidates"], shadow_api_ms=round(api_ms, 3),
                    predict_api_device_statistic_ms=round(predict_ms, 3), predict_api_host_statistic_ms=round(predict_host_ms, 3),
                    reference_published="2.65 s per predict() call (testing.ipynb:90, unnamed NVIDIA GPU, H2D included)"
                    if name == "testing" else None)
        if args.generic:
            class Plain(PathEmbedding):           # an overridden forward keeps the generic torch path
                def forward(self, x):
                    return super().forward(x)
            obj_g = PathShadowing(Plain(emb.kernel), RelativeMSE(), ds, PredictionContext(horizon=c["h"]))
            n_splits = 64 if name == "testing" else 8
            try:
                obj_g.shadow(x.numpy(), k=c["k"], n_splits=n_splits, cuda=True)
                t0 = time.perf_counter()
                dg, _, ig = obj_g.shadow(x.numpy(), k=c["k"], n_splits=n_splits, cuda=True)
                assert obj_g.last_path == "torch"
                line["generic_torch_on_device_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
                line["generic_max_rel_diff"] = float(np.max(np.abs(np.sort(dg, 1) - d) / d))
            except Exception as e:  # noqa: BLE001
                line["generic_torch_on_device_ms"] = f"failed: {type(e).__name__}: {str(e)[:80]}"
        print(json.dumps(line))


if __name__ == "__main__":
    main()
