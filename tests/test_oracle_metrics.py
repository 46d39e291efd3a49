"""Row f-4 on the CPU: the numpy restatement of the reference's offset / metric functions against
the reference-generated fixture (tests/golden/metrics.npz) and the reference tests' known answers,
and the host half of the product (utils/order_stats.py: order statistics from histograms, with
numpy's own quantile arithmetic) against numpy itself."""
import os

import numpy as np
import pytest

from oracle import host_oracle as H
from util import metric_inputs

from aind_exaspim_image_compression.utils import order_stats as OS

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PCTS = (0.0, 0.1, 1.0, 50.0, 99.9, 100.0)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "metrics.npz"))


def cases(seed):
    pu, pf, raw, target, fg = metric_inputs(seed)
    for pname, pred in (("u16", pu), ("f32", pf)):
        for rname, r in (("u16", raw), ("f32", raw.astype(np.float32))):
            yield f"s{seed}/{pname}_{rname}", pred, r, target, fg


@pytest.mark.parametrize("seed", [0, 1])
def test_metric_oracle_equals_reference_outputs(gold, seed):
    for tag, pred, raw, target, fg in cases(seed):
        ev = H.evaluate_example(pred, raw, target, fg)
        for k, v in ev.items():
            assert v == float(gold[f"{tag}/evaluate/{k}"]), (tag, k)
        np.testing.assert_array_equal(H.split_mae(pred, raw, fg), gold[f"{tag}/fb_mae"])
        assert H.mip_max_error(pred, raw) == float(gold[f"{tag}/mip_max_error"])
        assert H.false_bright_rate(pred, raw, fg, k=3.0) == float(gold[f"{tag}/false_bright_k3"])


@pytest.mark.parametrize("seed", [0, 1])
def test_offset_oracle_equals_reference_outputs(gold, seed):
    _, pf, raw, _, _ = metric_inputs(seed)
    for pct in PCTS:
        assert H.estimate_offset(raw, pct) == float(gold[f"s{seed}/estimate_offset/u16/{pct}"])
        assert H.estimate_offset(raw, pct, ignore_zeros=False) == float(
            gold[f"s{seed}/estimate_offset/u16_keepzeros/{pct}"])
        assert H.estimate_offset(pf - 125.0, pct) == float(gold[f"s{seed}/estimate_offset/f32/{pct}"])


def test_reference_known_answers(gold):
    """tests/test_transforms.py:120-129, tests/test_metrics.py:115-138 of the reference."""
    sample = np.arange(0, 101, dtype=np.float32)
    assert H.estimate_offset(sample, percentile=0) == 1.0
    assert H.estimate_offset(sample, percentile=100) == 100.0
    assert H.estimate_offset(sample, percentile=0, ignore_zeros=False) == 0.0
    np.testing.assert_array_equal(gold["kat/fb_mae"], [10.0, 20.0])
    assert H.split_mae(np.array([[10.0, 20.0]]), np.zeros((1, 2)), np.array([[True, False]])) == (10.0, 20.0)
    assert H.mip_max_error(np.array([1.0, 900.0]), np.array([0.0, 1000.0])) == 100.0 == float(gold["kat/mip"])
    raw = np.zeros(10)
    raw[0] = 5000.0
    fg = np.zeros(10, dtype=bool)
    fg[0] = True
    pred = np.zeros(10)
    pred[1] = 5000.0
    assert H.false_bright_rate(pred, raw, fg) == pytest.approx(1.0 / 9.0)


def test_ssim_oracle_properties():
    """The property the reference tests (tests/test_review_regressions.py:270-286): uint16 input
    gives the float result; plus identity -> 1 and the window placement scipy uses."""
    rng = np.random.default_rng(42)
    a = rng.integers(40000, 65000, size=(8, 8, 8), dtype=np.uint16)
    b = np.clip(a.astype(np.int32) + rng.integers(-1000, 1000, a.shape), 0, 65535).astype(np.uint16)
    assert H.ssim3d(a, b, window_size=3) == pytest.approx(
        H.ssim3d(a.astype(np.float64), b.astype(np.float64), window_size=3), abs=1e-12)
    assert H.ssim3d(a, a) == pytest.approx(1.0, abs=1e-9)
    from scipy.ndimage import uniform_filter
    x = np.arange(40, dtype=np.float64)
    got = uniform_filter(x, 16)
    idx = np.arange(40)[:, None] + np.arange(-8, 8)[None, :]           # window i-8 .. i+7
    idx = np.where(idx < 0, -idx - 1, idx)
    idx = np.where(idx >= 40, 2 * 40 - 1 - idx, idx)                   # reflect: d c b a | a b c d
    np.testing.assert_allclose(got, x[idx].mean(axis=1), rtol=0, atol=1e-12)


# ---- the product's host half: numpy's quantile arithmetic on order statistics -----------------------
@pytest.mark.parametrize("n", [1, 2, 3, 10, 101, 1000, 65537, 3_000_001])
def test_percentile_from_histogram_is_numpy_bit_for_bit(n):
    rng = np.random.default_rng(n)
    v = np.clip(rng.normal(300, 40, n), 0, 65535).astype(np.uint16)
    v[rng.random(n) < 0.1] = 0
    hist = np.bincount(v, minlength=65536)
    for dtype in (np.float32, np.float64, np.uint16):
        x = v.astype(dtype)
        for q in (0, 0.1, 1.0, 1, 37.5, 50, 99.9, 100, 0.001, 12.3456):
            for ignore in (False, True):
                xs = x[x > 0] if ignore and np.any(x > 0) else x
                want = np.percentile(xs, q)
                got = OS.percentile(OS.from_u16_hist(hist, ignore_zeros=ignore, dtype=dtype), q)
                assert want == got and np.asarray(want).dtype == np.asarray(got).dtype, (dtype, q)
    st = OS.from_u16_hist(hist, dtype=np.float64)
    x = v.astype(np.float64)
    med = OS.median(st)
    assert med == np.median(x)
    assert OS.median_abs_deviation(st, med) == np.median(np.abs(x - med))


def test_percentile_rejects_bad_input():
    st = OS.from_u16_hist(np.bincount([3, 4], minlength=65536))
    with pytest.raises(ValueError):
        OS.percentile(st, 101.0)
    with pytest.raises(ValueError):
        OS.percentile(OS.OrderStats(np.arange(4), np.zeros(4)), 50.0)


class _FakeCtx:
    """key_histogram of the C-ABI restated with numpy, to exercise the radix selection logic."""

    def __init__(self, data):
        self.data = np.asarray(data)

    def key_histogram(self, vol, dtype, n, digit, prefix=0, center=None):
        v = self.data.astype(np.float64)
        if center is not None:
            v = np.abs(v - center)
        bits = v.view(np.uint64)
        key = np.where(bits >> np.uint64(63), ~bits, bits | np.uint64(1 << 63))
        shift = np.uint64(48 - 16 * digit)
        if digit:
            key = key[(key >> (shift + np.uint64(16))) == np.uint64(prefix)]
        return np.bincount(((key >> shift) & np.uint64(0xFFFF)).astype(np.int64),
                           minlength=65536).astype(np.uint64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_radix_selection_logic(dtype):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.normal(0, 50, 5000), np.zeros(40), -np.zeros(7),
                        rng.uniform(1e-30, 1e30, 50)]).astype(dtype)
    srt = np.sort(x.astype(np.float64))
    st = OS.DeviceOrderStats(_FakeCtx(x), None, dtype, x.size)
    for k in (0, 1, 17, 2500, 2548, 5000, x.size - 1, -1):
        assert st.at(k) == srt[k]
    assert st.count_not_positive() == int(np.count_nonzero(x <= 0))
    for q in (0.0, 0.1, 50.0, 99.9, 100.0):
        assert OS.percentile(st, q) == np.percentile(x.astype(np.float64), q)
    med = OS.median(st)
    assert med == np.median(x.astype(np.float64))
    dev = OS.DeviceOrderStats(_FakeCtx(x), None, dtype, x.size, center=float(med))
    assert OS.median(dev) == np.median(np.abs(x.astype(np.float64) - med))
    pos = OS.Shifted(OS.DeviceOrderStats(_FakeCtx(x), None, dtype, x.size, dtype=np.float32),
                     st.count_not_positive())
    if dtype == np.float32:
        assert OS.percentile(pos, 1.0) == np.percentile(x[x > 0], 1.0)
