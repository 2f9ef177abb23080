"""Host-side mirror of the reference's plugin surface: names,
signatures, return types, error behaviour -- and the reference's own three test cells
(testing.ipynb: forward_topk prefix consistency, shadow() self-consistency)."""
from pathlib import Path

import numpy as np
import pytest
import torch

import shadowing_amd as sa
from _util import PREDICT_GOLDENS, SMALL_GOLDENS, assert_exact, load_golden, predict_case
from shadowing_amd import synthetic as syn
from shadowing_amd.path_shadowing import _dim_array, _numpy, _torch


def test_drop_in_import_paths():
    import shadowing
    from shadowing import (Foveal, Identity, PathShadowing, PredictionContext, RelativeMSE, Softmax,  # noqa: F401
                           realized_variance, select_cartesian_product)
    from shadowing.path_shadowing import PathDistance, PathEmbedding  # noqa: F401
    assert shadowing.PathShadowing is sa.PathShadowing
    # tutorial.ipynb:21-24 / testing.ipynb:134-136 import the figure helpers from the same package
    from shadowing import plot_closest, plot_shadow, plot_volatility  # noqa: F401


def test_plot_helpers_draw_without_scatspectra():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from shadowing import plot_closest, plot_shadow, plot_volatility
    rng = np.random.default_rng(0)
    present, paths, d = rng.standard_normal(20) * 0.01, rng.standard_normal((8, 1, 30)) * 0.01, np.linspace(0.3, 0.6, 8)
    plot_closest(present, paths, num_trajectories=5)
    plot_shadow(present, d, paths, eta=0.1)
    plot_volatility(present, np.full((2, 11), 0.2), [5, 10], distances=d, close_paths=paths, eta=0.1)
    assert len(plt.get_fignums()) == 3
    plt.close("all")


@pytest.mark.parametrize("name", SMALL_GOLDENS)
def test_generic_torch_path_reproduces_reference(name):
    """cuda=False runs the same torch formulation as the reference: same bits, same
    (even tie) order as the stored reference outputs."""
    g = load_golden(name)
    obj = sa.PathShadowing(sa.Identity(g["W"]), sa.RelativeMSE(), g["dataset"], sa.PredictionContext(g["h"]))
    q = g["queries"] if name != "k1_2d_dataset" else g["queries"][0]
    d, paths, idx = obj.shadow(q, k=g["k"], n_splits=g["n_splits"], cuda=False)
    assert d.dtype == np.float32 and paths.dtype == np.float32 and idx.dtype == np.int32
    assert np.array_equal(paths, g["paths"])
    assert_exact(d, idx, g["d"], g["idx"], name)


def test_forward_topk_prefix_consistency():
    """testing.ipynb cell 1: top-32 with 32 splits == first 32 of top-64 with 64 splits."""
    torch.manual_seed(0)
    dist = sa.RelativeMSE()
    x, y = torch.randn(8, 34), torch.randn(128, 96, 34)
    d32, i32 = dist.forward_topk(x, y, 32, 32)
    d64, i64 = dist.forward_topk(x, y, 64, 64)
    assert torch.equal(d32, d64[:, :32]) and torch.equal(i32, i64[:, :32])
    assert i32.dtype == torch.int64 and i32.shape == (8, 32, 2)
    # indices point at the distances they claim
    for b in range(8):
        r, t = i32[b, :, 0], i32[b, :, 1]
        assert torch.equal(dist(x[b][None, :], y[r, t]), d32[b])


def test_shadow_self_consistency_foveal():
    """testing.ipynb cell 2 (smaller): re-embedding the returned paths' in-context part
    reproduces the returned distances."""
    torch.manual_seed(1)
    emb = sa.Foveal(alpha=1.15, beta=0.9, max_context=126)
    ctx = sa.PredictionContext(horizon=50)
    ds = torch.randn(16, 1, 1024).numpy()
    x = torch.randn(4, 1, 126).numpy()
    obj = sa.PathShadowing(emb, sa.RelativeMSE(), ds, ctx)
    d, paths, idx = obj.shadow(x, k=64, cuda=False)
    assert paths.shape == (4, 64, 1, 176) and idx.shape == (4, 64, 2)
    hx = emb(torch.tensor(x))[:, 0, :]
    hp = emb(torch.tensor(ctx.select_in_context(paths)).reshape(-1, 1, 126))[:, 0, :].reshape(4, 64, -1)
    again = sa.RelativeMSE()(hx[:, None, :], hp).numpy()
    assert np.allclose(again, d, rtol=1e-4)
    assert np.all(np.diff(d, axis=1) >= 0)
    assert emb.dim == 34 and len(emb.slices) == 34 and emb.kernel.shape == (34, 1, 126)


def test_error_behaviour():
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), syn.dataset(4, 64, 0), sa.PredictionContext(20))
    with pytest.raises(Exception, match="same size as the context"):
        obj.shadow(np.zeros(19, np.float32), k=1)
    with pytest.raises(RuntimeError):          # k larger than the number of windows, as torch.topk in the reference
        obj.shadow(syn.single_query(20), k=1000)
    with pytest.raises(ValueError, match="Unrecognized averaging proba"):
        sa.PathShadowing.init_averaging_proba("median", np.zeros((1, 2, 1)), None)
    with pytest.raises(Exception, match="cannot be formatted"):
        _dim_array(np.zeros((1, 1, 1, 1)))
    assert isinstance(obj.context, sa.PredictionContext)
    assert sa.PathShadowing(sa.Identity(4), sa.RelativeMSE(), np.zeros((2, 16))).context.horizon is None


def test_helpers_and_contexts():
    assert _dim_array(np.zeros(5)).shape == (1, 1, 5) and _dim_array(np.zeros((3, 5))).shape == (3, 1, 5)
    assert _torch(np.zeros(3, np.float64)).dtype == torch.float32
    t = torch.ones(2)
    assert _torch(t) is t and isinstance(_numpy(t), np.ndarray)
    a, b, c = torch.arange(3), torch.arange(10, 14), torch.arange(20, 22)
    flat = torch.tensor([0, 5, 23, 7])
    assert torch.equal(sa.select_cartesian_product(flat, [a, b, c]), torch.cartesian_prod(a, b, c)[flat])
    x = np.arange(10.0)[None, :]
    p = sa.PredictionContext(3)
    assert p.get_out_times() == 3 and p.select_in_context(x).shape[-1] == 7 and np.array_equal(p.select_out_context(x)[0], [7, 8, 9])
    assert p.pad_context(torch.ones(2, 1, 4)).shape[-1] == 7
    none = sa.PredictionContext()
    assert none.get_out_times() == 0 and none.select_out_context(x) is x and none.pad_context(t) is t
    im = sa.ImputationContext((2, 3, 4))
    assert im.get_out_times() == 3 and np.array_equal(im.select_in_context(x)[0], [0, 1, 6, 7, 8, 9])
    assert np.array_equal(im.select_out_context(x)[0], [2, 3, 4, 5]) and im.slect_out_context(x).shape == (1, 4)
    assert torch.equal(im.pad_context(torch.arange(6.0)[None]), torch.tensor([[0, 1, 0, 0, 0, 2, 3, 4, 5.]]))
    cc = sa.CrossChannelContext(1)
    y = np.zeros((2, 3, 8))
    assert cc.select_in_context(y).shape == (2, 2, 8) and cc.select_out_context(y).shape == (2, 1, 8)
    assert cc.pad_context(torch.zeros(4, 2, 5)).shape == (4, 3, 5) and cc.get_out_times() == 0
    e = sa.Identity(6)
    assert e.d == 6 and e.kernel.shape == (6, 1, 6) and e(torch.randn(2, 1, 9)).shape == (2, 4, 6)
    assert e.adjust_to_context(p).kernel.shape == (6, 1, 9)
    xx = torch.randn(2, 1, 9)
    assert torch.equal(e(xx)[:, 1, :], xx[:, 0, 1:7])           # Identity conv == the window itself
    rv = sa.realized_variance(np.ones((2, 10)) * 0.1, [2, 5], vol=False)
    assert rv.shape == (2, 2) and np.allclose(rv, 0.01 * 252)


def test_predict_from_paths_with_a_fake_proba(monkeypatch):
    """The body of predict_from_paths (select_out_context -> proba.avg/std over axis 1)
    with a recording stand-in for the un-vendored scatspectra operators."""
    calls = {}

    class Fake:
        def avg(self, x, axis):
            calls["avg"] = (x.shape, axis)
            return x.mean(axis)

        def std(self, x, axis):
            calls["std"] = (x.shape, axis)
            return x.std(axis)

    obj = sa.PathShadowing(sa.Identity(4), sa.RelativeMSE(), np.zeros((2, 32)), sa.PredictionContext(3))
    monkeypatch.setattr(sa.PathShadowing, "init_averaging_proba", staticmethod(lambda n, d, e: (calls.setdefault("init", (n, d.shape, e)), Fake())[1]))
    paths = np.random.default_rng(0).standard_normal((5, 6, 1, 7)).astype(np.float32)
    m, s = obj.predict_from_paths(np.ones((5, 6), np.float32), paths, lambda f: (f ** 2).sum(-1), "softmax", 0.1)
    assert calls["init"] == ("softmax", (5, 6, 1), 0.1) and calls["avg"] == ((5, 6, 1), 1)
    assert np.allclose(m, (paths[..., -3:] ** 2).sum(-1).mean(1)) and m.shape == (5, 1) and s.shape == (5, 1)


@pytest.mark.parametrize("name", PREDICT_GOLDENS)
def test_predict_matches_the_reference_predict(name, monkeypatch):
    """predict() (shadow -> select_out_context -> proba.avg / std over the k paths, context splits concatenated) against
    the REFERENCE's own predict() run with the same known-arithmetic averaging classes (PS:245-252, 256-301): pins the
    slicing and axis conventions around the un-vendored classes against the reference, bit for bit on the host path."""
    obj, g = predict_case(name, sa, monkeypatch)
    Ts = [int(t) for t in g["Ts"]]
    m, s = obj.predict(g["queries"], int(g["k"]), lambda f: sa.realized_variance(f, Ts, vol=True), eta=g["eta"],
                       proba_name=str(g["proba_name"]), n_dataset_splits=int(g["n_dataset_splits"]),
                       n_context_splits=int(g["n_context_splits"]), cuda=False)
    assert m.shape == g["mean"].shape and s.shape == g["std"].shape
    assert np.array_equal(m, g["mean"]) and np.array_equal(s, g["std"])


def test_predict_runs_on_the_host_path():
    ds = syn.dataset(32, 256, 3)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds, sa.PredictionContext(10))
    q = syn.rolling_queries(4, 20, 4)
    m, s = obj.predict(q, k=16, to_predict=lambda f: sa.realized_variance(f, [5, 10], vol=False),
                       eta=0.5, n_context_splits=2, cuda=False)
    assert m.shape == (4, 1, 2) and s.shape == (4, 1, 2) and np.all(np.isfinite(m)) and np.all(s >= 0)
    mu, _ = obj.predict(q, k=16, to_predict=lambda f: f.mean(-1), proba_name="uniform", cuda=False)
    d, paths, _ = obj.shadow(q, k=16)
    assert np.allclose(mu, paths[..., -10:].mean(-1).mean(1))


def test_native_dispatch_rules():
    """Which configurations the HIP kernels take (host logic only, no device needed): Identity -> the
    plain scan, stock linear embeddings whose kernel fits LDS -> the embedded scan, everything else
    (overridden forward, other distances/contexts, several channels, oversized kernels) -> torch."""
    import torch
    import shadowing_amd as sa
    from shadowing_amd import _native

    def kind(emb, dist=None, ctx=None, x=None, y=None, k=8):
        obj = sa.PathShadowing(emb, dist or sa.RelativeMSE(), np.zeros((4, 1, 600), np.float32), ctx)
        K = emb.kernel.shape[-1]
        x = torch.zeros((2, 1, K)) if x is None else x
        y = torch.zeros((4, 1, 600)) if y is None else y
        return obj._native_kind(x, y, k)

    assert kind(sa.Identity(20)) == "identity"
    assert kind(sa.Foveal(1.15, 0.9, 126), ctx=sa.PredictionContext(252)) == "linear"
    assert kind(sa.PathEmbedding(torch.randn(5, 1, 23))) == "linear"
    assert kind(sa.PathEmbedding(torch.randn(128, 1, 256))) is None         # 128 x 256 taps do not fit LDS
    assert kind(sa.Foveal(1.15, 0.9, 252)) == "linear"                      # 39 x 252: the 8-wave instantiation
    assert _native.embedding_supported(32, 256) and not _native.embedding_supported(129, 4)

    class Squared(sa.PathEmbedding):
        def forward(self, x):
            return super().forward(x) ** 2

    class Padded(sa.PathEmbedding):
        def adjust_to_context(self, context):
            return self

    class L1(sa.PathDistance):
        def forward(self, x, y):
            return (x - y).abs().sum(-1)

    assert kind(Squared(torch.randn(3, 1, 10))) is None
    assert kind(Padded(torch.randn(3, 1, 10))) is None
    assert kind(sa.Identity(20), dist=L1()) is None
    assert kind(sa.Identity(20), ctx=sa.ImputationContext((8, 4, 8))) is None          # l + r != window length
    assert kind(sa.Identity(20), ctx=sa.ImputationContext((8, 5, 12))) == "padded"
    assert kind(sa.PathEmbedding(torch.randn(4, 1, 13)), ctx=sa.ImputationContext((6, 9, 7))) == "padded"
    assert kind(sa.Identity(20), ctx=sa.ImputationContext((8, 250, 12))) is None       # padded kernel beyond 256 taps
    assert kind(sa.Identity(20), ctx=sa.ImputationContext(None)) is None
    assert kind(Squared(torch.randn(3, 1, 10)), ctx=sa.ImputationContext((4, 3, 6))) is None
    assert kind(sa.Identity(20), y=torch.zeros((4, 2, 600))) is None        # two channels
    assert kind(sa.Identity(20), ctx=sa.CrossChannelContext(1), y=torch.zeros((4, 2, 600))) == "identity"
    assert kind(sa.Foveal(2.0, 0.5, 20), ctx=sa.CrossChannelContext(2), y=torch.zeros((4, 3, 600))) == "linear"
    assert kind(sa.Identity(20), ctx=sa.CrossChannelContext(2), y=torch.zeros((4, 2, 600))) is None   # channel count off
    assert kind(Squared(torch.randn(3, 1, 20)), ctx=sa.CrossChannelContext(1), y=torch.zeros((4, 2, 600))) is None
    assert kind(sa.Identity(20), k=_native.PSH_MAX_K + 1) is None
    assert kind(sa.Identity(20), x=torch.zeros((2, 1, 20), dtype=torch.float64)) is None


def test_host_path_rereads_the_ensemble_on_every_call():
    """cuda=False reads `dataset` afresh on every call, as the reference does (ref :205): nothing converted is kept,
    so an in-place edit of ONE row -- float32 (wrapped without a copy) or float64 (converted per call) -- is seen."""
    q = syn.gbm_log_returns((1, 10), 4)
    for dtype in (np.float32, np.float64):
        ds = syn.dataset(64, 300, 3).astype(dtype)
        obj = sa.PathShadowing(sa.Identity(10), sa.RelativeMSE(), ds, sa.PredictionContext(5))
        if dtype == np.float32:
            assert obj._dataset_tensor().data_ptr() == ds.__array_interface__["data"][0]      # no copy
        d1, _, i1 = obj.shadow(q, k=5)
        ds[40, 0, 100:110] = q[0]                                          # a surgical edit: one window of one row
        d2, _, i2 = obj.shadow(q, k=5)
        assert d2[0, 0] == 0.0 and tuple(i2[0, 0]) == (40, 100) and d1[0, 0] > 0.0
    assert obj._dataset_tensor().dtype == torch.float32


def test_resident_copy_policy():
    """What may stay in HBM between calls (cache="auto"): a torch tensor (version counter) or a read-only array;
    a writeable numpy array never; cache=True keeps anything until refresh(); cache=False nothing."""
    ds = syn.dataset(8, 100, 5)
    mk = lambda data, **kw: sa.PathShadowing(sa.Identity(10), sa.RelativeMSE(), data, sa.PredictionContext(5), **kw)
    assert not mk(ds)._may_keep_resident()
    ro = ds.copy(); ro.flags.writeable = False
    assert mk(ro)._may_keep_resident() and mk(torch.tensor(ds))._may_keep_resident()
    assert mk(ds, cache=True)._may_keep_resident() and not mk(ro, cache=False)._may_keep_resident()
    # a read-only VIEW of a writeable base can change through the base: never kept
    view = ds.view(); view.flags.writeable = False
    assert not mk(view)._may_keep_resident() and not mk(np.broadcast_to(ds[:1], ds.shape))._may_keep_resident()
    with pytest.raises(ValueError):
        mk(ds, cache="yes")
    obj = mk(ro)
    assert obj.last_path is None                                           # readable before the first call
    assert obj._dataset_tensor().data_ptr() == ro.__array_interface__["data"][0]     # read-only arrays are wrapped too


def test_writeable_numpy_ensemble_warns_once_and_predict_shares_one_upload():
    """The drop-in default (a writeable numpy array, what the tutorial passes): cuda=True re-uploads it per call like the
    reference -- but says so ONCE, with the remedy, and the context splits of one predict() share one upload.  (The upload
    itself is exercised here with the host standing in for the device: _resident_dataset only moves tensors.)"""
    import warnings
    ds = syn.dataset(8, 100, 5)
    obj = sa.PathShadowing(sa.Identity(10), sa.RelativeMSE(), ds, sa.PredictionContext(5))
    y = obj._dataset_tensor()
    cpu = torch.device("cpu")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        a = obj._resident_dataset(y, cpu)
        b = obj._resident_dataset(y, cpu)
    assert len(w) == 1 and issubclass(w[0].category, RuntimeWarning)
    assert "cache=True" in str(w[0].message) and "refresh()" in str(w[0].message) and "every call" in str(w[0].message)
    assert a is not b or a.data_ptr() == y.data_ptr()                       # (per call: nothing is kept between calls)
    assert obj._resident is None
    # inside one predict(): the first call's upload serves the later context splits; afterwards nothing is kept
    uploads = []
    orig = torch.Tensor.to

    def counting_to(self, *a_, **k_):
        if self.shape == y.shape:
            uploads.append(1)
        return orig(self, *a_, **k_)
    obj._predict_scope = (obj.dataset, None)
    torch.Tensor.to = counting_to
    try:
        u1 = obj._resident_dataset(y, cpu)
        u2 = obj._resident_dataset(y, cpu)
        u3 = obj._resident_dataset(y, cpu)
    finally:
        torch.Tensor.to = orig
        obj._predict_scope = None
    assert len(uploads) == 1 and u1 is u2 is u3
    # an edit of the array between two predict() calls is seen by the second (the scope ends with the call)
    obj._predict_scope = (obj.dataset, None)
    first = obj._resident_dataset(y, cpu).clone()
    obj._predict_scope = None
    ds[0, 0, 0] += 1.0
    obj._predict_scope = (obj.dataset, None)
    second = obj._resident_dataset(obj._dataset_tensor(), cpu)
    obj._predict_scope = None
    assert np.float32(second[0, 0, 0]) == np.float32(first[0, 0, 0]) + np.float32(1.0)
    # predict() opens and closes the scope itself (host path: cuda=False leaves it alone)
    mean, std = obj.predict(syn.gbm_log_returns((4, 10), 9), k=5, to_predict=lambda p: p.mean(-1), proba_name="uniform", n_context_splits=2)
    assert obj._predict_scope is None and mean.shape[0] == 4
    # cache=True / a tensor / a read-only array: no warning
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sa.PathShadowing(sa.Identity(10), sa.RelativeMSE(), ds, sa.PredictionContext(5), cache=True)._resident_dataset(y, cpu)
        sa.PathShadowing(sa.Identity(10), sa.RelativeMSE(), torch.tensor(ds), sa.PredictionContext(5))._resident_dataset(torch.tensor(ds), cpu)
    assert not w
    with pytest.raises(ValueError):
        sa.PathShadowing(sa.Identity(10), sa.RelativeMSE(), ds, sa.PredictionContext(5), hint="always")


def test_realized_variance_takes_torch_tensors():
    x = syn.gbm_log_returns((3, 4, 30), 5)
    for vol in (False, True):
        a = sa.realized_variance(x, [2, 7, 30], vol=vol)
        b = sa.realized_variance(torch.tensor(x), [2, 7, 30], vol=vol)
        assert isinstance(b, torch.Tensor) and b.shape == (3, 4, 3)
        np.testing.assert_allclose(b.numpy(), a, rtol=1e-6)


@pytest.mark.parametrize("name", ["forward_topk_d34", "forward_topk_d5", "forward_topk_d126"])
def test_forward_topk_matches_reference_golden(name, oracle_mod):
    """The generic forward_topk (flat top-k + decoded indices instead of the reference's host-side
    itertools.product table) against the reference's own output; and the observation the HIP path rests on: the
    same numbers come out of the scan over N one-window paths (oracle, T' = 1: contiguous 8-lane reduce)."""
    z = np.load(Path(__file__).resolve().parent / "golden" / f"{name}.npz")
    x, y, k = torch.tensor(z["x"]), torch.tensor(z["y"]), int(z["k"])
    d, idx = sa.RelativeMSE().forward_topk(x, y, k, n_splits=int(z["n_splits"]))
    assert idx.dtype == torch.int64 and idx.shape == z["idx"].shape
    assert np.array_equal(d.numpy().view(np.uint32), z["d"].view(np.uint32))
    for b in range(x.shape[0]):
        assert {tuple(v) for v in idx[b].numpy()} == {tuple(v) for v in z["idx"][b]}
    od, oi = oracle_mod.scan_topk(z["y"].reshape(-1, 1, y.shape[-1]).copy(), z["x"], k, h=0)
    assert np.array_equal(od.view(np.uint32), np.sort(z["d"], 1).view(np.uint32)) and oi[..., 1].max() == 0
    T2 = y.shape[1]
    for b in range(x.shape[0]):
        assert {(int(f) // T2, int(f) % T2) for f in oi[b, :, 0]} == {tuple(v) for v in z["idx"][b]}


def test_moment_weights_keeps_the_averaging_class_authoritative():
    """path_shadowing.moment_weights: weights are taken for the device reduction only when the class's OWN avg / std are
    the weighted moments of the weights it exposes (checked on a probe); anything else reduces on the host itself."""
    from shadowing_amd.path_shadowing import moment_weights
    from shadowing_amd.averaging import Softmax, Uniform
    d = np.random.default_rng(0).random((3, 50, 1)) + 0.3
    w = moment_weights(Softmax(d, 0.2), 3, 50)
    assert isinstance(w, np.ndarray) and w.shape == (3, 50) and w.dtype == np.float64 and np.allclose(w.sum(axis=1), 1.0)
    assert moment_weights(Uniform(), 3, 50) is True
    assert moment_weights(Softmax(d, 0.2), 3, 49) is None          # shape mismatch

    class NoWeights:
        def avg(self, x, axis=1):
            return x.mean(axis=1)

        def std(self, x, axis=1):
            return x.std(axis=1)

    assert moment_weights(NoWeights(), 3, 50) is None

    class OtherStd(Softmax):                                     # exposes weights, reduces differently
        def std(self, x, axis=0):
            return np.abs(x).max(axis=axis)

    assert moment_weights(OtherStd(d, 0.2), 3, 50) is None

    class Unnormalised(Softmax):                                 # weights as given: nothing is renormalised
        def __init__(self, dd):
            super().__init__(dd, 0.2)
            self.weights = self.weights * 3.0

    w3 = moment_weights(Unnormalised(d), 3, 50)
    assert isinstance(w3, np.ndarray) and np.allclose(w3.sum(axis=1), 3.0)
