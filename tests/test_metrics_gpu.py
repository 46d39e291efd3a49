"""GPU parity of row f-4 (background offset + quality metrics on device) through the C-ABI:
histogram / radix-selection / masked-error / SSIM kernels against numpy, the host oracle and the
reference-generated fixture tests/golden/metrics.npz."""
import os

import numpy as np
import pytest

from oracle import host_oracle as H
from util import metric_inputs

from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.machine_learning import metrics as M
from aind_exaspim_image_compression.machine_learning import transforms as T
from aind_exaspim_image_compression.utils import img_util as IU
from aind_exaspim_image_compression.utils import order_stats as OS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PCTS = (0.0, 0.1, 1.0, 50.0, 99.9, 100.0)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "metrics.npz"))


# ---- histogram kernel ---------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 8, 4097, 57344, 57345, 1_000_003, 9_000_001])
def test_u16_histogram_exact(ctx, n):
    rng = np.random.default_rng(n)
    if n % 2:
        v = rng.integers(0, 65536, n, dtype=np.uint16)               # every bin, worst case for the flush
    else:
        v = np.clip(rng.normal(200, 30, n), 0, 65535).astype(np.uint16)   # hot bins
    buf = ctx.to_device(v)
    try:
        np.testing.assert_array_equal(ctx.u16_histogram(buf, n), np.bincount(v, minlength=65536))
    finally:
        buf.free()


def test_u16_histogram_unaligned_and_saturating_bins(ctx):
    """A base pointer that is not 16-byte aligned takes the scalar path; one value repeated far
    beyond 65535 times must not carry between the packed 16-bit LDS counters."""
    v = np.full(3_000_000, 65534, dtype=np.uint16)
    v[::3] = 65535
    v[1::1000] = 0
    buf = ctx.to_device(v)
    try:
        np.testing.assert_array_equal(ctx.u16_histogram(buf, v.size), np.bincount(v, minlength=65536))
        np.testing.assert_array_equal(ctx.u16_histogram(buf.ptr + 2, v.size - 1),
                                      np.bincount(v[1:], minlength=65536))
    finally:
        buf.free()


@pytest.mark.parametrize("dtype", [np.uint16, np.float32, np.float64])
def test_radix_order_statistics_exact(ctx, dtype):
    rng = np.random.default_rng(5)
    n = 300_001
    if dtype == np.uint16:
        x = rng.integers(0, 5000, n).astype(dtype)
    else:
        x = np.concatenate([rng.normal(100, 400, n - 50), np.zeros(30), -np.zeros(20)]).astype(dtype)
    buf = ctx.to_device(x)
    try:
        st = OS.DeviceOrderStats(ctx, buf, dtype, n)
        srt = np.sort(x.astype(np.float64))
        for k in (0, 1, 1234, n // 2, n - 2, n - 1):
            assert st.at(k) == srt[k]
        assert st.count_not_positive() == int(np.count_nonzero(x <= 0))
        for q in PCTS:
            assert OS.percentile(st, q) == np.percentile(x.astype(np.float64), q)
        med = OS.median(st)
        dev = OS.DeviceOrderStats(ctx, buf, dtype, n, center=float(med))
        assert OS.median(dev) == np.median(np.abs(x.astype(np.float64) - med))
    finally:
        buf.free()


# ---- estimate_offset ----------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1])
def test_estimate_offset_equals_reference_fixture(gold, seed):
    _, pf, raw, _, _ = metric_inputs(seed)
    for pct in PCTS:
        assert T.estimate_offset(raw, pct) == float(gold[f"s{seed}/estimate_offset/u16/{pct}"])
        assert T.estimate_offset(raw, pct, ignore_zeros=False) == float(
            gold[f"s{seed}/estimate_offset/u16_keepzeros/{pct}"])
        assert T.estimate_offset(pf - 125.0, pct) == float(gold[f"s{seed}/estimate_offset/f32/{pct}"])


def test_estimate_offset_reference_cases():
    """reference tests/test_transforms.py:120-129, plus the all-zero sample."""
    sample = np.arange(0, 101, dtype=np.float32)
    assert T.estimate_offset(sample, percentile=0) == 1.0
    assert T.estimate_offset(sample, percentile=100) == 100.0
    assert T.estimate_offset(sample, percentile=0, ignore_zeros=False) == 0.0
    assert T.estimate_offset(np.zeros(100, np.uint16)) == 0.0
    assert T.estimate_offset(np.zeros(100, np.float32)) == 0.0
    with pytest.raises(ValueError):
        T.estimate_offset(np.zeros(0, np.uint16))


def test_background_offset_statistics(ctx):
    rng = np.random.default_rng(11)
    vol = np.clip(rng.normal(150, 20, (50, 60, 70)), 0, 65535).astype(np.uint16)
    vol[:5] = 0
    want = H.background_offset_statistics(vol, 0.1)
    assert T.background_offset_statistics(vol, 0.1) == want
    empty = T.background_offset_statistics(np.zeros((4, 4, 4), np.uint16), 0.1)
    assert np.isnan(empty["offset"]) and np.isnan(empty["median"]) and empty["zero_fraction"] == 1.0
    assert empty["offset_all_voxels"] == 0.0


def test_estimate_offset_on_a_resident_volume(ctx):
    vol = metric_inputs(0, shape=(96, 100, 104))[2]
    buf = ctx.to_device(vol)
    try:
        assert T.estimate_offset_device(ctx, buf, vol.size, 0.1) == H.estimate_offset(vol, 0.1)
    finally:
        buf.free()


# ---- metrics -------------------------------------------------------------------------------------------
def cases(seed):
    pu, pf, raw, target, fg = metric_inputs(seed)
    for pname, pred in (("u16", pu), ("f32", pf)):
        for rname, r in (("u16", raw), ("f32", raw.astype(np.float32))):
            yield f"s{seed}/{pname}_{rname}", pred, r, target, fg


@pytest.mark.parametrize("seed", [0, 1])
def test_metrics_equal_reference_fixture(gold, seed):
    """Everything that is an order statistic, a count, a maximum or a sum of integers is exact;
    float absolute-error sums differ by fp64 summation order only (tolerance 1e-13 relative)."""
    for tag, pred, raw, target, fg in cases(seed):
        ev = M.evaluate_example(pred, raw, target, fg)
        for k in ("top_pct_error", "top_pct_preservation", "mip_max_error", "false_bright_rate"):
            assert ev[k] == float(gold[f"{tag}/evaluate/{k}"]), (tag, k)
        for k in ("fg_mae", "bg_mae"):
            assert ev[k] == pytest.approx(float(gold[f"{tag}/evaluate/{k}"]), rel=1e-13), (tag, k)
        fb = M.foreground_background_mae(pred, raw, fg)
        if pred.dtype == np.uint16:
            np.testing.assert_array_equal(fb, gold[f"{tag}/fb_mae"])     # integer sums: exact
        else:
            np.testing.assert_allclose(fb, gold[f"{tag}/fb_mae"], rtol=1e-13)
        assert M.mip_max_error(pred, raw) == float(gold[f"{tag}/mip_max_error"])
        assert M.false_bright_rate(pred, raw, fg, k=3.0) == float(gold[f"{tag}/false_bright_k3"])


def test_metrics_reference_test_cases():
    """reference tests/test_metrics.py:115-170."""
    fg_mae, bg_mae = M.foreground_background_mae(np.array([[10.0, 20.0]]), np.zeros((1, 2)),
                                                 np.array([[True, False]]))
    assert (fg_mae, bg_mae) == (10.0, 20.0)
    assert M.mip_max_error(np.array([1.0, 900.0]), np.array([0.0, 1000.0])) == 100.0
    raw = np.zeros(10)
    raw[0] = 5000.0
    fg = np.zeros(10, dtype=bool)
    fg[0] = True
    pred = np.zeros(10)
    pred[1] = 5000.0
    assert M.false_bright_rate(pred, raw, fg) == pytest.approx(1.0 / 9.0)
    assert M.false_bright_rate(pred, raw, np.ones(10, dtype=bool)) == 0.0
    raw = np.zeros((16, 16, 16), dtype=np.float32)
    raw[4:12, 4:12, 4:12] = 60000
    fg = raw > 1000
    ev = M.evaluate_example(raw, raw, raw, fg)
    assert set(ev) == {"fg_mae", "bg_mae", "top_pct_error", "top_pct_preservation",
                       "mip_max_error", "false_bright_rate"}
    assert ev["fg_mae"] == 0.0 and ev["mip_max_error"] == 0.0
    assert ev["top_pct_preservation"] == pytest.approx(1.0, abs=1e-5)
    ev = M.evaluate_example(raw * 0.5, raw, raw, fg)
    assert ev["top_pct_preservation"] < 1.0 and ev["mip_max_error"] > 0.0
    assert ev == H.evaluate_example(raw * 0.5, raw, raw, fg)
    m = {"fg_mae": 10.0, "bg_mae": 5.0, "top_pct_error": 20.0}
    assert M.checkpoint_score(m, 3.0) == 10.0 + 0.2 * 5.0 + 0.5 * 20.0
    assert M.checkpoint_score(m, 3.0, {"fg_mae": 1.0, "cratio": 2.0}) == 10.0 - 6.0


def test_mae_lmax(ctx):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 4000, (33, 47, 51)).astype(np.uint16)
    b = rng.integers(0, 4000, (33, 47, 51)).astype(np.uint16)
    assert IU.compute_mae(a, b) == H.compute_mae(a, b)
    assert IU.compute_lmax(a, b) == H.compute_lmax(a, b)
    af = a.astype(np.float32) + 0.25
    assert IU.compute_mae(af, b) == pytest.approx(H.compute_mae(af, b), rel=1e-13)
    assert IU.compute_lmax(af, b.astype(np.int64)) == H.compute_lmax(af, b)


# ---- SSIM ------------------------------------------------------------------------------------------------
SSIM_CASES = [
    ((8, 8, 8), 3), ((8, 8, 8), 16), ((20, 33, 70), 16), ((64, 64, 64), 16), ((37, 18, 129), 7),
    ((40, 40, 40), 5), ((16, 100, 65), 32), ((5, 3, 2), 4), ((70, 17, 64), 1),
]


@pytest.mark.parametrize("shape,window", SSIM_CASES)
def test_ssim_equals_oracle(shape, window):
    """uint16 input: every local moment is an exact integer / window^3, so only the order of the
    final mean (and, for windows that are not powers of two, scipy's per-axis division) differs:
    1e-12 relative.  float64 input: running fp64 box sums, 1e-9."""
    rng = np.random.default_rng(sum(shape) + window)
    a = np.clip(rng.normal(300, 60, shape), 0, 65535).astype(np.uint16)
    a[tuple(s // 2 for s in shape)] = 40000
    b = np.clip(a.astype(np.float64) + rng.normal(0, 25, shape), 0, 65535).astype(np.uint16)
    want = H.ssim3d(a, b, window_size=window)
    assert IU.ssim3D(a, b, window_size=window) == pytest.approx(want, rel=1e-12, abs=1e-14)
    want = H.ssim3d(a, b, data_range=np.max(a), window_size=window)   # evaluate.py:105 calling form
    assert IU.ssim3D(a, b, data_range=np.max(a), window_size=window) == pytest.approx(want, rel=1e-12)
    af, bf = a.astype(np.float64) * 0.37, b.astype(np.float32) * 0.37
    want = H.ssim3d(af, bf, window_size=window)
    assert IU.ssim3D(af, bf, window_size=window) == pytest.approx(want, rel=1e-9)


def test_ssim_reference_test_case_and_errors():
    """reference tests/test_review_regressions.py:270-286."""
    rng = np.random.default_rng(42)
    a = rng.integers(40000, 65000, size=(8, 8, 8), dtype=np.uint16)
    b = np.clip(a.astype(np.int32) + rng.integers(-1000, 1000, a.shape), 0, 65535).astype(np.uint16)
    ri = IU.ssim3D(a, b, window_size=3)
    rf = IU.ssim3D(a.astype(np.float64), b.astype(np.float64), window_size=3)
    assert ri == pytest.approx(rf, abs=1e-12)
    assert ri == pytest.approx(H.ssim3d(a, b, window_size=3), abs=1e-12)
    assert IU.ssim3D(a, a) == pytest.approx(1.0, abs=1e-9)
    with pytest.raises(ValueError):
        IU.ssim3D(a, b[:4])
    with pytest.raises(ValueError):
        IU.ssim3D(a, b, window_size=33)


def test_metrics_are_deterministic(ctx):
    pu, pf, raw, target, fg = metric_inputs(3, shape=(64, 72, 80))
    r1 = (IU.ssim3D(pu, raw), M.evaluate_example(pf, raw, target, fg))
    r2 = (IU.ssim3D(pu, raw), M.evaluate_example(pf, raw, target, fg))
    assert r1 == r2


def test_missing_library_or_bad_arguments_fail_loudly(ctx):
    with pytest.raises(ValueError):
        ctx.key_histogram(0, np.float32, 10, 5)            # NULL volume
    buf = ctx.to_device(np.zeros(16, np.float32))
    try:
        with pytest.raises(ValueError):
            ctx.key_histogram(buf, np.float32, 16, 4)      # digit out of range
        with pytest.raises(ValueError):
            ctx.key_histogram(buf, np.float32, 16, 1, prefix=1 << 16)
    finally:
        buf.free()
    assert isinstance(_native.context(0), _native.Context)
