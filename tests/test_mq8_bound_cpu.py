"""The 8-bit rejection test of the batched scan (scan_mq8_kernel) EMULATED in numpy and fuzzed on the CPU.

What is restated here, step for step and rounding for rounding (shadowing_amd/csrc/: mq_prep_kernel psh_scan.hip, the
"scan_mq8_kernel" part of threshold_kernel psh_select.hip, the per-segment set-up and the group loop of scan_mq8_kernel
psh_scan.hip):
    batch      : s0 = max|x| / 127, x^ = rint(x / s0) clamped, E0^2 = max_q ||x - s0 x^||^2, NX0 = max_q ||x||^2 (rounded up)
    per query  : beta, P (rounded up), L = (||x^||_1 / 2 + 3.5)(1 + 2^-16), k1 (rounded down) from tau, the f16 scale sc
    per segment: y~ = sc y, lm = max|y~|, inv_sy = 127 / max(lm, 2^-6), y^ = round-to-nearest-even(y~ inv_sy) as bytes,
                 the f16 squares (y~^2)^, the window energies ny = sum of W of them, C_w = (int) min(ny kC, 2^30) with
                 kC = min(inv_sy k1, 2048), thr = (int) clamp(fma(P, inv_sy, L)), keep_all when lm > 128 / NaN
    per window : reject iff  C_w - sum_j x^_j y^_{t+j}  >  thr
The property (what "rigorous" means; the GPU side of it is tests/test_gpu_admitted_set.py): NO window whose exact fp32 chain
acc = sum_j (x_j - y_{t+j})^2 lies below tau is ever rejected -- for any data, any batch, any level.  hypothesis drives the
shapes that stress the bound: amplitudes over decades inside one batch, segments far quieter than the batch's scale, spikes
the scale never saw, subnormal squares, W = 8..25, levels from the 10th to the 10^5-th smallest acc."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

f32 = np.float32
SEG = 1024


def quant(x, inv_s0):
    """mq8_quant (psh_device.h): (int) rintf(x * inv_s0), clamped to [-127, 127] (x * inv_s0 is an fp32 product)."""
    return np.clip(np.rint((x.astype(f32) * f32(inv_s0)).astype(f32)), -127, 127).astype(np.int32)


def up(v: float) -> np.float32:
    """(float) v rounded UP to the next float when the conversion rounded down (threshold_kernel's `if ((double)Pf < P)`)."""
    f = f32(v)
    if float(f) < v:
        f = np.nextafter(f, f32(np.inf))
    return f


def pow2_scale(maxabs: np.float32) -> np.float32:
    """2^sexp that puts `maxabs` into [4, 8): sexp = 3 - (exponent - 126) of the bit pattern (threshold_kernel)."""
    bits = int(np.asarray(maxabs, f32).view(np.uint32))
    e = ((bits >> 23) & 255) - 126
    sexp = 3 - e
    if not (-60 <= sexp <= 60 and bits >= 0x00800000):
        return f32(0)
    return f32(2.0 ** sexp)


def batch_constants(q):
    """mq_prep_kernel's last block: (xmax, inv_s0, E0^2, NX0) -- UNSCALED."""
    xmax = f32(np.abs(q).max())
    if not (xmax > 0 and np.isfinite(xmax)):
        return xmax, f32(0), f32(0), f32(0)
    inv_s0 = f32(127.0) / xmax
    s0 = 1.0 / float(inv_s0)
    e2m, nxm = f32(0), f32(0)
    for x in q:
        xp = np.zeros(25, f32); xp[:len(x)] = x
        r = xp.astype(np.float64) - s0 * quant(xp, inv_s0).astype(np.float64)
        e2m = max(e2m, up(float((r * r).sum())))
        nxm = max(nxm, up(float((xp.astype(np.float64) ** 2).sum())))
    return xmax, inv_s0, f32(e2m), f32(nxm)


def query_constants(x, tau, sc, xmax, inv_s0, E0, NX0):
    """threshold_kernel, "scan_mq8_kernel": (P, L, k1, -x^) of one query, or None when the test is not armed."""
    sc = float(sc)
    armed = sc > 0 and xmax > 0 and np.isfinite(xmax) and np.isfinite(float(xmax) * sc)
    if not armed or not np.isfinite(tau):
        return None
    s_x = sc / float(inv_s0)
    E2, NX = float(E0) * sc * sc, float(NX0) * sc * sc
    if not (NX > 0 and np.isfinite(E2)):
        return None
    beta = min(max(np.sqrt(E2 / NX), 1.0 / 4096.0), 0.25)
    xp = np.zeros(25, f32); xp[:len(x)] = x
    xs = (xp * f32(sc)).astype(f32).astype(np.float64)
    xq = quant(xp, inv_s0)
    nxs, l1 = float((xs * xs).sum()), float(np.abs(xq).sum())
    taus = float(tau) * sc * sc
    Theta = taus * (1.0 + 1.0 / 65536.0) - nxs * (1.0 - 1e-12)
    P = (Theta + E2 * (1.0 + 1e-6) / beta) / (2.0 * s_x)
    P = P * (1.0 + 1.0 / 1048576.0) if P > 0 else P * (1.0 - 1.0 / 1048576.0)
    Pf = up(P)
    Lf = f32(f32(0.5 * l1 + 3.5) * f32(1.0 + 1.0 / 65536.0))
    k1 = f32((1.0 - beta) * (1.0 - 1.0 / 256.0) / (2.0 * s_x) * (1.0 - 1.0 / 1048576.0))
    if not (np.isfinite(Pf) and k1 > 0 and np.isfinite(k1)):
        return None
    return Pf, Lf, k1, -xq[:len(x)]


def rne_bytes(v, inv_sy):
    """y^ of a segment: the fma(v, inv_sy, 1.5 * 2^23) trick -- the EXACT product rounded once, to nearest even."""
    p = v.astype(np.float64) * float(inv_sy)                 # exact: 24 x 24 bits
    return np.rint(p).astype(np.int32)                       # numpy rint is round-half-even; |p| <= 127 here


def segment_rejects(yseg, W, sc, consts):
    """Reject mask (per query, per window) of one wave segment of SEG windows; yseg holds SEG + W - 1 values."""
    v = (yseg.astype(f32) * f32(sc)).astype(f32)
    v2 = (v * v).astype(f32)
    lm = f32(np.abs(v).max()) if np.isfinite(v).all() else f32(np.nan)
    nan = not np.isfinite(v2.sum(dtype=np.float64)) and np.isnan(v2).any()
    k1 = max((c[2] for c in consts if c is not None), default=f32(0))
    keep_all = (not (lm <= 128.0)) or nan or not (sc > 0) or not (k1 > 0)
    nwin = len(yseg) - W + 1
    if keep_all:
        return np.zeros((len(consts), nwin), bool)
    inv_sy = f32(127.0) / max(lm, f32(0.015625))
    yq = rne_bytes(v, inv_sy)
    assert np.abs(yq).max() <= 127
    e16 = v2.astype(np.float16).astype(f32)                  # (y~^2)^: round-to-nearest-even conversion
    # window energies: the MFMA adds the W exact f16 values in an fp32 accumulator (any order: within a few ulps, the bound's
    # 2^-8 covers it 10^4 times over); here in double, rounded once
    cs = np.concatenate([[0.0], np.cumsum(e16.astype(np.float64))])
    ny = (cs[W:W + nwin] - cs[:nwin]).astype(f32)
    kC = min(f32(inv_sy * k1), f32(2048.0))
    cw = np.minimum((ny * kC).astype(f32), f32(1073741824.0)).astype(np.int64)     # (int) truncates; values are >= 0
    rej = np.zeros((len(consts), nwin), bool)
    idx = np.arange(nwin)[:, None] + np.arange(W)[None, :]
    for qi, c in enumerate(consts):
        if c is None:
            continue                                          # P = +inf in the kernel: keeps everything
        Pf, Lf, _, xneg = c
        t = np.float64(Pf) * np.float64(inv_sy) + np.float64(Lf)          # the kernel's fma: one rounding
        thr = int(np.clip(f32(t), f32(-2147483520.0), f32(2147483520.0)))  # (int) truncates towards zero
        prod = (yq[idx].astype(np.int64) * xneg.astype(np.int64)[None, :]).sum(axis=1)
        rej[qi] = (cw + prod) > thr
    return rej


def exact_acc(yrow, x):
    """The reference's chain per window: D = fl(x_j - y_{t+j}), acc = fma(D, D, acc) (path_distance.py:62-65)."""
    W = len(x)
    n = len(yrow) - W + 1
    acc = np.zeros(n, f32)
    for j in range(W):
        D = (f32(x[j]) - yrow[j:j + n].astype(f32)).astype(f32).astype(np.float64)
        acc = (D * D + acc.astype(np.float64)).astype(f32)  # D^2 is exact in double; one rounding of the sum (a rare double rounding is
    return acc                                               # 1 ulp of acc: the bound's margins are 2^-16)


@settings(max_examples=120, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(seed=st.integers(0, 2 ** 31 - 1), W=st.sampled_from([8, 13, 17, 20, 20, 25]), B=st.integers(1, 6),
       spread=st.sampled_from([1.0, 1.0, 3.0, 30.0, 1e3]), qscale=st.sampled_from([1.0, 1.0, 1e-3, 1e-5, 1e3, 1e-1]),
       scale=st.sampled_from([1.0, 1.0, 1e-12, 1e12, 1e-17, 1e-3]), kind=st.sampled_from(["plain", "spike", "quiet", "planted", "heavy", "zeros", "unsampled_loud"]),
       depth=st.sampled_from([10, 100, 1000, 5000]))
def test_8bit_rejection_never_rejects_a_window_below_the_level(seed, W, B, spread, qscale, scale, kind, depth):
    rng = np.random.default_rng(seed)
    nrow = 4
    y = (rng.standard_normal((nrow, SEG + W - 1)) * 0.0126).astype(f32)
    q = (rng.standard_normal((B, W)) * 0.0126).astype(f32)
    q *= rng.uniform(1.0, spread, (B, 1)).astype(f32)
    q *= f32(qscale)
    if kind == "spike":
        y[0, rng.integers(0, y.shape[1], 3)] *= f32(10.0 ** rng.integers(1, 6))
    elif kind == "quiet":
        y[1] *= f32(1e-5); y[2, :400] *= f32(1e-4)
    elif kind == "planted":
        for b in range(B):
            t = int(rng.integers(0, SEG - W))
            y[b % nrow, t:t + W] = q[b] * (1 + f32(10.0 ** -rng.integers(1, 6)) * rng.standard_normal(W).astype(f32))
        y[3, 5:5 + W] = q[0]
    elif kind == "heavy":
        y = (0.01 * rng.standard_t(2.5, size=y.shape)).astype(f32)
    elif kind == "zeros":
        y[0] = 0; y[1, ::2] = 0; q[0, ::3] = 0
    y = (y * f32(scale)).astype(f32); q = (q * f32(scale)).astype(f32)
    # the f16 scale: the largest |value| of the SAMPLED rows and of the batch into [4, 8) -- "unsampled_loud": a row the sample
    # never saw is 300x louder (beyond 128 after scaling: keep_all has to catch it)
    sampled = np.abs(y[:2]).max()
    if kind == "unsampled_loud":
        y[3] *= f32(300.0)
    maxabs = f32(max(sampled, np.abs(q).max()))
    if not (maxabs > 0 and np.isfinite(maxabs)):
        return
    sc = pow2_scale(maxabs)
    xmax, inv_s0, E0, NX0 = batch_constants(q)
    accs = [np.stack([exact_acc(y[r], q[b]) for r in range(nrow)]) for b in range(B)]
    consts, taus = [], []
    for b in range(B):
        a = np.sort(accs[b][np.isfinite(accs[b])].ravel())
        tau = f32(a[min(depth, len(a) - 1)])
        taus.append(tau)
        consts.append(query_constants(q[b], tau, sc, xmax, inv_s0, E0, NX0) if tau > 0 else None)
    n_rejected = 0
    for r in range(nrow):
        rej = segment_rejects(y[r], W, sc, consts)
        for b in range(B):
            below = accs[b][r] < taus[b]
            false_rejects = np.nonzero(below & rej[b])[0]
            assert false_rejects.size == 0, (f"query {b} row {r}: windows {false_rejects[:5].tolist()} with acc "
                                             f"{accs[b][r][false_rejects[:5]].tolist()} < tau {taus[b]!r} were REJECTED by the 8-bit test")
            n_rejected += int(rej[b].sum())
    # (no assertion on the filter's power here -- tools/ and the GPU tests measure that; this is about rigor only)


def test_the_emulation_rejects_most_windows_at_the_benchmarks_shape():
    """Sanity of the emulation itself: on the benchmark's kind of data (i.i.d. Gaussian log-returns, rolling queries, a level
    ~1000 deep) the emulated test must reject the bulk -- an emulation that kept everything would make the property vacuous."""
    rng = np.random.default_rng(0)
    W, B = 20, 4
    y = (rng.standard_normal((8, SEG + W - 1)) * 0.0126).astype(f32)
    path = (rng.standard_normal(W + B - 1) * 0.0126).astype(f32)
    q = np.stack([path[b:b + W] for b in range(B)])
    sc = pow2_scale(f32(max(np.abs(y).max(), np.abs(q).max())))
    xmax, inv_s0, E0, NX0 = batch_constants(q)
    kept = total = 0
    for b in range(B):
        acc = np.stack([exact_acc(y[r], q[b]) for r in range(8)])
        tau = f32(np.sort(acc.ravel())[8])
        c = query_constants(q[b], tau, sc, xmax, inv_s0, E0, NX0)
        assert c is not None
        for r in range(8):
            rej = segment_rejects(y[r], W, sc, [c])[0]
            assert not (rej & (acc[r] < tau)).any()
            kept += int((~rej).sum()); total += rej.size
    assert kept < 0.05 * total, (kept, total)


def test_a_weakened_bound_is_caught(monkeypatch):
    """The fuzz has teeth: with the level side of the test shrunk by 15 % (the bound's own slack is ~9 % of the level at the
    benchmark's sizes) the same examples produce false rejects."""
    import sys
    me = sys.modules[__name__]
    orig = me.query_constants

    def weak(*a, **k):
        c = orig(*a, **k)
        if c is None:
            return c
        P, L, k1, x = c
        return (f32(P * 0.85 if P > 0 else P * 1.15), L, k1, x)
    monkeypatch.setattr(me, "query_constants", weak)
    with pytest.raises(AssertionError, match="REJECTED by the 8-bit test"):
        test_8bit_rejection_never_rejects_a_window_below_the_level()
