"""SURVEY.md 8f row 3, second half: predict_from_paths()'s reductions on the device -- psh_weighted_moments (avg / std over
the k paths, reference path_shadowing.py:245-252) and psh_realized_variance (shadowing/statistics.py:5-16) -- against numpy
in the reference's own expressions, and predict(cuda=True, device_predict=True) end to end against the reference's predict()."""
import numpy as np
import pytest
import torch

from _util import syn

pytestmark = pytest.mark.gpu


def _ref_rv(x, Ts, vol):
    # the reference's expression (statistics.py:12-16), float32 as numpy runs it
    x2 = x ** 2
    r = np.stack([x2[..., :T].mean(-1) for T in Ts], -1) * 252
    return r ** 0.5 if vol else r


@pytest.mark.parametrize("shape,offset,Ts,vol", [((6, 300, 1, 272), 20, [5, 10, 20, 252], True),
                                                  ((2, 17, 3, 90), 40, [1, 2, 50, 64, 3], False),
                                                  ((1, 1, 1, 378), 126, list(range(1, 64)), True),
                                                  ((4, 9, 1, 30), 0, [30, 45], False)])
def test_realized_variance_kernel_equals_numpy(hip_device, shape, offset, Ts, vol):
    """On the out-context VIEW of gathered paths (rows a constant stride apart, no copy) and on a contiguous tensor; a
    maturity beyond the row clips as numpy's slice does.  Squares in fp32, sums in double: within 2 ulp of numpy's
    pairwise fp32 mean."""
    from shadowing_amd import _native, realized_variance
    g = np.random.default_rng(7)
    paths = (g.standard_normal(shape) * 0.0126).astype(np.float32)
    paths[0, 0, 0, offset:] *= 1e3                                      # a loud row
    want = _ref_rv(paths[..., offset:], Ts, vol)
    pt = torch.as_tensor(paths).to(hip_device)
    view = pt[..., offset:]
    n_rows, stride = _native._uniform_rows(view)
    assert n_rows == int(np.prod(shape[:-1])) and (n_rows == 1 or stride == shape[-1])
    got = realized_variance(view, Ts, vol)
    assert got.is_cuda and got.dtype == torch.float32 and tuple(got.shape) == want.shape
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=3e-7)
    got2 = realized_variance(view.contiguous(), Ts, vol)               # contiguous rows: stride == length
    assert torch.equal(got, got2)
    # a layout that is not rows a constant stride apart goes through torch ops (same numbers to fp32 rounding)
    odd = pt[:, ::2, :, offset:] if shape[1] > 2 else pt.transpose(0, 1)[..., offset:]
    if _native._uniform_rows(odd) is None:
        np.testing.assert_allclose(realized_variance(odd, Ts, vol).cpu().numpy(), _ref_rv(odd.cpu().numpy(), Ts, vol), rtol=2e-6)


@pytest.mark.parametrize("B,k,tail", [(6, 8192, (3,)), (1, 10000, (1,)), (3, 257, (2, 5)), (2, 64, (1500,)), (5, 1, (4,))])
@pytest.mark.parametrize("uniform", [False, True])
def test_weighted_moments_kernel_equals_numpy_float64(hip_device, B, k, tail, uniform):
    """avg / std over axis 1 exactly as averaging.py / scatspectra-style classes define them (float64 sums)."""
    from shadowing_amd import _native
    g = np.random.default_rng(B * 1000 + k)
    v = (g.standard_normal((B, k) + tail) * 0.2 + 0.15).astype(np.float32)
    w = None
    if not uniform:
        w = np.exp(-g.random((B, k)) * 30.0)
        w /= w.sum(axis=1, keepdims=True)
    wf = np.full((B, k), 1.0 / k) if w is None else w
    wb = wf.reshape((B, k) + (1,) * len(tail))
    x = v.astype(np.float64)
    m0 = (wb * x).sum(axis=1)
    s0 = np.sqrt((wb * (x - np.expand_dims(m0, 1)) ** 2).sum(axis=1))
    mean, std = _native.weighted_moments(torch.as_tensor(v).to(hip_device), None if w is None else torch.as_tensor(w).to(hip_device))
    assert mean.dtype == torch.float64 and tuple(mean.shape) == (B,) + tail
    np.testing.assert_allclose(mean.cpu().numpy(), m0, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(std.cpu().numpy(), s0, rtol=1e-10, atol=1e-14)
    with pytest.raises(ValueError):
        _native.weighted_moments(torch.as_tensor(v).to(hip_device), torch.zeros((B, k + 1), dtype=torch.float64, device=hip_device))


def test_predict_reduces_on_the_device_and_hands_other_classes_the_statistic(hip_device, monkeypatch):
    """predict(cuda=True, device_predict=True): with the stock averaging classes only the (B, k) distances and the (B, m)
    moments cross PCIe (`last_predict_reduction == "device"`), same numbers as the host path; a class whose avg / std are
    NOT the weighted moments of exposed weights keeps reducing on the host, by its own rules."""
    import shadowing_amd as sa
    ds = syn.dataset(2048, 1024, 81)
    x = syn.gbm_log_returns((4, 20), 82)
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), ds, sa.PredictionContext(horizon=40), cache=True)
    Ts = [5, 20, 40]

    def stat(p):
        return sa.realized_variance(p, Ts, vol=True)

    for proba, eta in (("softmax", 0.05), ("uniform", None)):
        m, s = obj.predict(x, k=512, to_predict=stat, eta=eta, proba_name=proba, cuda=True, device_predict=True)
        assert obj.last_path == "hip" and obj.last_predict_reduction == "device"
        d, paths, _ = obj.shadow(x, k=512, cuda=True)
        m0, s0 = obj.predict_from_paths(d, paths, stat, proba, eta)
        assert m.shape == m0.shape == (4, 1, 3) and m.dtype == np.float64
        np.testing.assert_allclose(m, m0, rtol=2e-6)
        np.testing.assert_allclose(s, s0, rtol=2e-5, atol=1e-12)

    class MaxDev:                                            # exposes weights, but its std is not the second moment
        def __init__(self, d):
            w = np.exp(-np.asarray(d, np.float64)[:, :, 0])
            self.weights = w / w.sum(axis=1, keepdims=True)

        def avg(self, v, axis=1):
            return (self.weights.reshape(self.weights.shape + (1,) * (np.ndim(v) - 2)) * np.asarray(v, np.float64)).sum(axis=1)

        def std(self, v, axis=1):
            return np.abs(np.asarray(v, np.float64) - np.expand_dims(self.avg(v), 1)).max(axis=1)

    monkeypatch.setattr(sa.PathShadowing, "init_averaging_proba", staticmethod(lambda name, d, eta: MaxDev(d)))
    m, s = obj.predict(x, k=512, to_predict=stat, eta=None, proba_name="softmax", cuda=True, device_predict=True)
    assert obj.last_predict_reduction == "host"
    d, paths, _ = obj.shadow(x, k=512, cuda=True)
    pr = MaxDev(d[:, :, None])
    v = stat(obj.context.select_out_context(paths))
    np.testing.assert_allclose(m, pr.avg(v), rtol=2e-6)
    np.testing.assert_allclose(s, pr.std(v), rtol=2e-5)
