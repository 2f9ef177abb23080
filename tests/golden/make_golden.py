#!/usr/bin/env python
"""Generate the golden vectors in this directory by running the READ-ONLY
reference (/root/reference, RudyMorel/shadowing @ 2024-12-20) on CPU.

Run in the build container only (the reference does not travel):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--big]

`scatspectra` (an un-vendored dependency the scan never touches when the
dataset is an array) is replaced by a names-only stub so that
`import shadowing` succeeds.  Every fixture stores the inputs (or, for large
datasets, the generator seed + SHA-256 of the bytes), the query norms torch
computed, and the reference's raw outputs of
PathShadowing.shadow(..., cuda=False): distances, indices and gathered paths.
Nothing but data is written: no reference source is copied.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.dont_write_bytecode = True


def load_reference():
    stub = types.ModuleType("scatspectra")
    for name in ("TimeSeriesDataset", "Softmax", "Uniform", "DiscreteProba", "PriceData", "windows"):
        setattr(stub, name, type(name, (), {}))
    sys.modules["scatspectra"] = stub
    sys.path.insert(0, "/root/reference")
    import shadowing  # noqa: F401  (the reference)
    sys.path.pop(0)
    return sys.modules["shadowing"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also the cfg-2 / cfg-3 sized cases (~2 min)")
    ap.add_argument("--embedded", action="store_true",
                    help="ONLY the linear-embedding cases (Foveal / user kernels); the Identity fixtures are left alone")
    ap.add_argument("--topk", action="store_true", help="ONLY the PathDistance.forward_topk cases")
    ap.add_argument("--batched", action="store_true",
                    help="ONLY the configs[2]-shaped cases with more queries than one query chunk of the batched scan")
    ap.add_argument("--edge", action="store_true",
                    help="ONLY the embedded one-window-per-row cases (T == K + h)")
    ap.add_argument("--predict", action="store_true",
                    help="ONLY the predict() cases: the reference's own predict() (PS:256-301) with averaging classes of "
                         "known arithmetic (tests/_known_proba.py) injected where un-vendored scatspectra's would be")
    ap.add_argument("--cross", action="store_true",
                    help="ONLY the CrossChannelContext cases (multi-channel ensemble, scan on channel 0)")
    ap.add_argument("--nan", action="store_true", help="ONLY the NaN / inf-in-the-ensemble cases of the Identity scan")
    ap.add_argument("--sharded", action="store_true",
                    help="ONLY BASELINE configs[3]: the ensembles bench.py --gpus N scans (rank blocks dataset(32768,4096,seed=g), "
                         "g < N) for N = 2, 4, 8, the reference run on each WHOLE ensemble (~10 min, ~10 GB)")
    args = ap.parse_args()

    ref = load_reference()
    import torch
    # synthetic generator of the build (pure numpy), loaded by path so that the
    # build's own `shadowing` alias package never shadows the reference here
    import importlib.util
    spec = importlib.util.spec_from_file_location("psh_synthetic", REPO / "shadowing_amd" / "synthetic.py")
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)

    def run(name, ds, q, W, h, k, n_splits, store_dataset, meta=None):
        obj = ref.PathShadowing(ref.Identity(W), ref.RelativeMSE(), ds, ref.PredictionContext(horizon=h))
        t0 = time.time()
        d, paths, idx = obj.shadow(q, k=k, n_splits=n_splits, cuda=False)
        dt = time.time() - t0
        q2 = np.atleast_2d(q).astype(np.float32)
        xn = torch.tensor(q2).norm(dim=-1).numpy()
        out = dict(queries=q2, xn=xn, d=d, idx=idx, paths=paths,
                   W=W, h=-1 if h is None else h, k=k, n_splits=n_splits,
                   dataset_sha256=syn.sha256(ds), dataset_shape=np.array(ds.shape))
        if store_dataset:
            out["dataset"] = ds
        m = dict(meta or {})
        m.update(reference_seconds=round(dt, 3), numpy=np.__version__, torch=torch.__version__)
        out["meta"] = json.dumps(m)
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(f"{name}: d{d.shape} idx{idx.shape} paths{paths.shape} ref {dt:.2f}s")

    def run_embedded(name, emb, ds, q, h, k, n_splits, store_dataset, meta=None, ctx=None):
        """A PathEmbedding with a (d,1,K) kernel in front of RelativeMSE: besides the outputs,
        the fixture keeps the kernel and the reference's embedded queries hx = embedding(x)."""
        obj = ref.PathShadowing(emb, ref.RelativeMSE(), ds, ctx if ctx is not None else ref.PredictionContext(horizon=h))
        t0 = time.time()
        d, paths, idx = obj.shadow(q, k=k, n_splits=n_splits, cuda=False)
        dt = time.time() - t0
        q2 = np.atleast_2d(q).astype(np.float32)
        hx = emb(torch.tensor(q2)[:, None, :])[:, 0, :]
        out = dict(queries=q2, kernel=emb.kernel[:, 0, :].numpy(), hx=hx.numpy(), hxnorm=hx.norm(dim=-1).numpy(),
                   d=d, idx=idx, paths=paths, h=-1 if h is None else h, k=k, n_splits=n_splits,
                   dataset_sha256=syn.sha256(ds), dataset_shape=np.array(ds.shape))
        if ctx is not None:      # the context's zero taps are part of the scanning kernel (no trailing horizon)
            out["kernel_padded"] = ctx.pad_context(emb.kernel)[:, 0, :].numpy()
            out["portion"] = np.array(ctx.portion)
        if store_dataset:
            out["dataset"] = ds
        else:   # generated ensemble, large k: keep the head of the gathered paths only
            out["paths"] = paths[:, :32]
        m = dict(meta or {})
        m.update(reference_seconds=round(dt, 3), numpy=np.__version__, torch=torch.__version__)
        out["meta"] = json.dumps(m)
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(f"{name}: kernel{tuple(emb.kernel.shape)} d{d.shape} idx{idx.shape} paths{paths.shape} ref {dt:.2f}s")

    def run_cross(name, emb, ds, q, oc, k, n_splits):
        """CrossChannelContext(oc) (path_embedding.py:91-114): the ensemble has 1 + oc channels, the query only the
        first; pad_context gives the scanning kernel zero taps on the other channels."""
        obj = ref.PathShadowing(emb, ref.RelativeMSE(), ds, ref.CrossChannelContext(oc))
        t0 = time.time()
        d, paths, idx = obj.shadow(q, k=k, n_splits=n_splits, cuda=False)
        dt = time.time() - t0
        hx = emb(torch.tensor(q))[:, 0, :]
        out = dict(queries=q, kernel=emb.kernel[:, 0, :].numpy(), hx=hx.numpy(), d=d, idx=idx, paths=paths, h=-1, k=k,
                   n_splits=n_splits, out_context_channels=oc, dataset=ds, dataset_sha256=syn.sha256(ds),
                   dataset_shape=np.array(ds.shape),
                   meta=json.dumps(dict(reference_seconds=round(dt, 3), numpy=np.__version__, torch=torch.__version__)))
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(f"{name}: dataset{ds.shape} d{d.shape} idx{idx.shape} paths{paths.shape} ref {dt:.2f}s")

    if args.nan:
        # NaN / +-inf planted in the ENSEMBLE: torch.topk(largest=False) ranks NaN distances last (PS:165), so windows that
        # touch a NaN never enter the top-k while k finite ones exist; an infinite sample makes its windows' distances +inf
        ds = syn.dataset(96, 700, 50)
        g = np.random.default_rng(51)
        for r, t in zip(g.integers(0, 96, 40), g.integers(0, 700, 40)):
            ds[r, 0, t] = np.nan
        for r, t in zip(g.integers(0, 96, 6), g.integers(0, 700, 6)):
            ds[r, 0, t] = np.inf if (r + t) % 2 else -np.inf
        ds[7, 0, :] = np.nan                                   # a whole row
        run("nan_in_ensemble_B1", ds, syn.single_query(20, 52), 20, 20, 64, 3, True)
        run("nan_in_ensemble_B3", ds, syn.rolling_queries(3, 20, 53), 20, 20, 64, 2, True)
        run("nan_in_ensemble_B9", ds, syn.rolling_queries(9, 20, 54), 20, 20, 48, 1, True)
        return

    if args.sharded:
        # what `bench.py --gpus N` scans: rank g holds dataset(32768, 4096, seed=g) as rows [g*32768, (g+1)*32768); the
        # reference sees the N blocks as ONE ensemble (it has no multi-GPU path: PS:170-173 is the merge being restated)
        blocks = [syn.dataset(32768, 4096, g) for g in range(8)]
        q = syn.single_query(20, 1)
        out = dict(queries=np.atleast_2d(q).astype(np.float32), W=20, h=20, k=1024, rows_per_rank=32768, T=4096,
                   block_sha256=np.array([syn.sha256(b) for b in blocks]))
        secs = {}
        for N in (2, 4, 8):
            ds = np.concatenate(blocks[:N], axis=0)
            obj = ref.PathShadowing(ref.Identity(20), ref.RelativeMSE(), ds, ref.PredictionContext(horizon=20))
            t0 = time.time()
            d, paths, idx = obj.shadow(q, k=1024, n_splits=64 * N, cuda=False)
            secs[N] = round(time.time() - t0, 1)
            out[f"d_N{N}"], out[f"idx_N{N}"] = d, idx
            print(f"cfg4 N={N}: R={ds.shape[0]} d{d.shape} idx{idx.shape} ref {secs[N]}s", flush=True)
            del obj, ds
        out["meta"] = json.dumps(dict(gen="concatenate([dataset(32768,4096,g) for g in range(N)])", qgen="single_query(20,1)",
                                      reference_seconds=secs, numpy=np.__version__, torch=torch.__version__))
        np.savez_compressed(HERE / "cfg4_R262144.npz", **out)
        return

    if args.predict:
        spec2 = importlib.util.spec_from_file_location("psh_known_proba", REPO / "tests" / "_known_proba.py")
        kp = importlib.util.module_from_spec(spec2)
        spec2.loader.exec_module(kp)
        ps_mod = sys.modules["shadowing.path_shadowing.path_shadowing"]     # PS:9 bound the names at import: rebind them there
        ps_mod.Softmax, ps_mod.Uniform, ps_mod.DiscreteProba = kp.Softmax, kp.Uniform, kp.DiscreteProba
        Ts = [5, 10, 20]

        def rv(x):                                          # the tutorial's statistic (statistics.py:5-16), vol=True
            return ref.realized_variance(x, Ts, vol=True)

        def run_predict(name, emb, ds, q, h, k, eta, proba_name, n_ds, n_ctx):
            obj = ref.PathShadowing(emb, ref.RelativeMSE(), ds, ref.PredictionContext(horizon=h))
            mean, std = obj.predict(q, k, rv, eta=eta, proba_name=proba_name, n_dataset_splits=n_ds, n_context_splits=n_ctx)
            d, paths, idx = obj.shadow(q, k=k, n_splits=n_ds, cuda=False)
            out = dict(queries=np.asarray(q, dtype=np.float32), dataset=ds, h=h, k=k, eta=-1.0 if eta is None else eta,
                       proba_name=proba_name, n_dataset_splits=n_ds, n_context_splits=n_ctx, Ts=np.array(Ts), mean=mean, std=std,
                       d=d, idx=idx, dataset_sha256=syn.sha256(ds),
                       meta=json.dumps(dict(numpy=np.__version__, torch=torch.__version__, proba="tests/_known_proba.py")))
            if isinstance(emb, ref.Foveal):
                out.update(foveal=np.array([1.15, 0.9, emb.kernel.shape[-1]]))
            np.savez_compressed(HERE / f"{name}.npz", **out)
            print(f"{name}: mean{mean.shape} std{std.shape} {mean.dtype}")

        run_predict("predict_identity_softmax", ref.Identity(20), syn.dataset(96, 400, 90), syn.gbm_log_returns((6, 20), 91),
                    20, 48, 0.05, "softmax", 2, 3)
        run_predict("predict_identity_uniform", ref.Identity(20), syn.dataset(96, 400, 90), syn.gbm_log_returns((4, 1, 20), 92),
                    20, 32, None, "uniform", 1, 1)
        run_predict("predict_foveal_softmax", ref.Foveal(1.15, 0.9, 64), syn.dataset(80, 500, 93), syn.gbm_log_returns((4, 64), 94),
                    20, 40, 0.1, "softmax", 1, 2)
        return

    if args.batched:
        # BASELINE.json configs[2] in small: more rolling query dates than one query chunk (112) of the batched scan
        run("cfg3_rolling_B128_R256", syn.dataset(256, 1024, 60), syn.rolling_queries(128, 20, 61), 20, 20, 64, 4, False,
            dict(gen="dataset(256,1024,60)", qgen="rolling_queries(128,20,61)"))
        run("cfg3_rolling_B130_R1024", syn.dataset(1024, 1024, 62), syn.rolling_queries(130, 20, 63), 20, 20, 100, 16, False,
            dict(gen="dataset(1024,1024,62)", qgen="rolling_queries(130,20,63)"))
        return

    if args.edge:
        # one window per row behind a linear embedding (T == K + h): conv1d's output collapses to (S, 1, d) and
        # RelativeMSE's norm becomes the contiguous reduce over d (path_embedding.py:129-132, path_distance.py:65)
        g = torch.Generator().manual_seed(80)
        run_embedded("foveal_one_window_rows", ref.Foveal(alpha=1.3, beta=0.8, max_context=30), syn.dataset(400, 37, 81),
                     syn.gbm_log_returns((3, 30), 82), 7, 50, 1, True)
        run_embedded("user_kernel_d20_one_window_rows", ref.PathEmbedding(torch.randn(20, 1, 16, generator=g)),
                     syn.dataset(300, 16, 83), syn.gbm_log_returns((2, 16), 84), None, 40, 2, True)
        run_embedded("user_kernel_d9_one_window_rows", ref.PathEmbedding(torch.randn(9, 1, 12, generator=g)),
                     syn.dataset(200, 15, 85), syn.gbm_log_returns((2, 12), 86), 3, 30, 1, True)
        return

    if args.topk:
        # PathDistance.forward_topk (path_distance.py:10-49) on a pre-embedded y (B2, T2, d)
        g = torch.Generator().manual_seed(70)
        for name, (B1, B2, T2, d, k, ns) in {"forward_topk_d34": (3, 40, 30, 34, 50, 2), "forward_topk_d5": (2, 100, 7, 5, 20, 4),
                                             "forward_topk_d126": (2, 30, 20, 126, 10, 1)}.items():
            x = torch.randn(B1, d, generator=g) * 0.02
            y = torch.randn(B2, T2, d, generator=g) * 0.02
            dr, ir = ref.RelativeMSE().forward_topk(x, y, k, n_splits=ns)
            np.savez_compressed(HERE / f"{name}.npz", x=x.numpy(), y=y.numpy(), d=dr.numpy(), idx=ir.numpy(), k=k, n_splits=ns,
                                meta=json.dumps(dict(numpy=np.__version__, torch=torch.__version__)))
            print(f"{name}: x{tuple(x.shape)} y{tuple(y.shape)} d{tuple(dr.shape)} idx{tuple(ir.shape)} {ir.dtype}")
        return

    if args.cross:
        def multi(R, T, C, seed):
            return np.ascontiguousarray(np.concatenate([syn.dataset(R, T, seed + c) for c in range(C)], axis=1))
        run_cross("crosschannel_identity_C2", ref.Identity(20), multi(70, 400, 2, 50), syn.gbm_log_returns((3, 1, 20), 52), 1, 48, 2)
        run_cross("crosschannel_foveal_C3", ref.Foveal(alpha=2.0, beta=0.5, max_context=32), multi(40, 300, 3, 53),
                  syn.gbm_log_returns((2, 1, 32), 56), 2, 30, 1)
        return

    if args.embedded:
        # the tutorial's embedding (tutorial.ipynb cell 8): Foveal(1.15, 0.9, 126)
        fov = ref.Foveal(alpha=1.15, beta=0.9, max_context=126)
        run_embedded("foveal_tutorial_small", fov, syn.dataset(64, 600, 30), syn.gbm_log_returns((3, 126), 31),
                     20, 64, 2, True)
        run_embedded("foveal_a2_hNone", ref.Foveal(alpha=2.0, beta=0.5, max_context=64), syn.dataset(48, 400, 32),
                     syn.gbm_log_returns((2, 64), 33), None, 32, 1, True)
        # a user kernel through the base class: dense random taps, K not a multiple of 4
        g = torch.Generator().manual_seed(34)
        user = ref.PathEmbedding(torch.randn(5, 1, 23, generator=g))
        run_embedded("user_kernel_d5_K23", user, syn.dataset(40, 256, 35), syn.gbm_log_returns((4, 23), 36),
                     7, 20, 1, True)
        # ragged rows + more queries than one accumulator group
        run_embedded("foveal_ragged_B7", ref.Foveal(alpha=1.3, beta=1.0, max_context=40), syn.dataset(33, 1100, 37),
                     syn.gbm_log_returns((7, 40), 38), 3, 100, 3, True)
        # BASELINE.json configs[4] in small: a real wavelet filter bank over a W = 252 window, batched queries
        wav = ref.PathEmbedding(torch.tensor(syn.wavelet_bank(5, 252))[:, None, :])
        run_embedded("wavelet_W252_rolling", wav, syn.dataset(48, 1500, 41), syn.rolling_queries(4, 252, 42),
                     20, 128, 2, True)
        # ImputationContext: l known samples, a gap of c to impute, r known samples (path_embedding.py:59-88)
        run_embedded("imputation_identity_8_5_12", ref.Identity(20), syn.dataset(60, 500, 43), syn.gbm_log_returns((3, 20), 44),
                     None, 40, 2, True, ctx=ref.ImputationContext((8, 5, 12)))
        run_embedded("imputation_user_kernel_6_9_7", ref.PathEmbedding(torch.randn(4, 1, 13, generator=g)), syn.dataset(50, 300, 45),
                     syn.gbm_log_returns((2, 13), 46), None, 25, 1, True, ctx=ref.ImputationContext((6, 9, 7)))
        # Foveal + ImputationContext on an ensemble large enough for the sampled path (the suffix-rows scan with a gap)
        run_embedded("imputation_foveal_18_6_12", ref.Foveal(alpha=1.3, beta=0.8, max_context=30), syn.dataset(600, 700, 47),
                     syn.gbm_log_returns((3, 30), 48), None, 200, 2, False, dict(gen="dataset(600,700,47)"),
                     ctx=ref.ImputationContext((18, 6, 12)))
        if args.big:
            # the tutorial's shape (k = 8192, horizon 252) on a generated ensemble
            run_embedded("foveal_tutorial_R1024", fov, syn.dataset(1024, 2048, 39), syn.gbm_log_returns((2, 126), 40),
                         252, 8192, 8, False, dict(gen="dataset(1024,2048,39)", qgen="gbm_log_returns((2,126),40)"))
        return

    # --- BASELINE.json configs[0]: the reference's own CPU-runnable case ----------------
    ds = syn.dataset(256, 1024, 0)
    q = syn.single_query(20, 1)
    run("cfg1_h20", ds, q, 20, 20, 64, 1, False, dict(gen="dataset(256,1024,0)", qgen="single_query(20,1)"))
    run("cfg1_hNone", ds, q, 20, None, 64, 1, False, dict(gen="dataset(256,1024,0)", qgen="single_query(20,1)"))

    # --- multi-query, multi-split ------------------------------------------------------
    ds = syn.dataset(96, 512, 10)
    run("multiquery_splits", ds, syn.rolling_queries(5, 20, 11), 20, 20, 48, 4, True)
    # --- remainder split (S % n_splits != 0), odd W -------------------------------------
    ds = syn.dataset(100, 384, 12)
    run("remainder_split_W12", ds, syn.gbm_log_returns((2, 12), 13), 12, 7, 32, 3, True)
    run("oddW33_h11", syn.dataset(64, 300, 14), syn.gbm_log_returns((2, 33), 15), 33, 11, 50, 2, True)
    run("W7_h0", syn.dataset(64, 300, 16), syn.gbm_log_returns((3, 7), 17), 7, 0, 16, 1, True)
    # --- exact ties: every path appears twice --------------------------------------------
    half = syn.dataset(48, 256, 18)
    run("duplicated_paths", np.concatenate([half, half], 0), syn.gbm_log_returns((2, 20), 19), 20, 20, 33, 1, True)
    # --- zero query: every distance is +inf ----------------------------------------------
    run("zero_query", syn.dataset(16, 128, 20), np.zeros((1, 20), np.float32), 20, 20, 8, 1, True)
    # --- query cut out of the dataset: an exact 0 distance --------------------------------
    ds = syn.dataset(32, 256, 21)
    run("self_match", ds, ds[3, 0, 100:120].copy(), 20, 20, 8, 1, True)
    # --- one window per row (T == W + h): the numerator's reduction order changes ---------
    run("single_window_rows", syn.dataset(300, 25, 22), syn.gbm_log_returns((2, 20), 23), 20, 5, 40, 1, True)
    # --- 2-D dataset (R, T) and 1-D query, k = 1 -----------------------------------------
    run("k1_2d_dataset", syn.dataset(40, 200, 24)[:, 0, :], syn.single_query(20, 25), 20, 20, 1, 1, True)

    if args.big:
        # BASELINE.json configs[1] size: R=32768, T=4096, W=20, k=1024, single query
        ds = syn.dataset(32768, 4096, 0)
        run("cfg2_R32768", ds, syn.single_query(20, 1), 20, 20, 1024, 64, False,
            dict(gen="dataset(32768,4096,0)", qgen="single_query(20,1)"))
        # configs[2] shape at a CPU-tractable size: rolling-window queries
        ds = syn.dataset(2048, 4096, 2)
        run("cfg3_rolling_R2048", ds, syn.rolling_queries(4, 20, 3), 20, 20, 1024, 32, False,
            dict(gen="dataset(2048,4096,2)", qgen="rolling_queries(4,20,3)"))


if __name__ == "__main__":
    main()
