"""Count-space quality metrics, reduced on the GPU.

Drop-in for the scoring half of the reference's ``machine_learning/metrics.py`` (lines 306-450):
``foreground_background_mae``, ``mip_max_error``, ``false_bright_rate``, ``evaluate_example``,
``checkpoint_score`` with the reference's signatures and return values.  The reference makes
float64 copies of every image and reduces them with numpy; here each image is uploaded once (or
is already a device buffer), one HIP kernel accumulates the masked absolute-error sums, maxima
and the bright-voxel count (``exabm4d_masked_error_stats_dev``), and the percentiles / median /
MAD come from device histograms (``utils/order_stats.py``).  Integer-valued inputs give the
reference's numbers exactly; float inputs differ only by fp64 summation order.

The mask builders of the reference module (segmentation / skeleton / coherence masks, lines
32-303) belong to the training data pipeline and are out of scope.
"""
import numpy as np

from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.utils import order_stats

# reference metrics.py:24-29
DEFAULT_CHECKPOINT_WEIGHTS = {
    "fg_mae": 1.0,
    "bg_mae": 0.2,
    "top_pct_error": 0.5,
    "cratio": 0.0,
}


class DeviceImage:
    """An image in HBM with the element type the kernels read (uint16, float32 or float64).

    uint16 and float32 arrays are uploaded as they are; every other dtype is widened to float64,
    which is what the reference converts everything to."""

    def __init__(self, arr, ctx=None):
        arr = np.asarray(arr)
        self.orig_dtype = arr.dtype
        if arr.dtype == np.bool_:
            arr = arr.astype(np.uint16)
        if arr.dtype not in (np.uint16, np.float32, np.float64):
            arr = arr.astype(np.float64)
        self.ctx = ctx or _native.context()
        self.shape = arr.shape
        self.dtype = arr.dtype
        self.n = int(arr.size)
        if self.n == 0:
            raise ValueError("empty image")
        self.buf = self.ctx.to_device(np.ascontiguousarray(arr).reshape(-1))
        self._stats = None

    def free(self):
        self.buf.free()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def order_stats(self):
        """Exact order statistics of the image as float64 values."""
        if self._stats is None:
            if self.dtype == np.uint16:
                hist = self.ctx.u16_histogram(self.buf, self.n)
                self._stats = order_stats.from_u16_hist(hist, dtype=np.float64)
            else:
                self._stats = order_stats.DeviceOrderStats(self.ctx, self.buf, self.dtype, self.n)
        return self._stats

    def median_abs_deviation(self, center):
        """``np.median(np.abs(x - center))`` of the float64-widened image."""
        if self.dtype == np.uint16:
            return order_stats.median_abs_deviation(self.order_stats(), center)
        dev = order_stats.DeviceOrderStats(self.ctx, self.buf, self.dtype, self.n,
                                           center=float(center))
        return order_stats.median(dev)


def _mask_buffer(ctx, fg_mask, n):
    fg = np.ascontiguousarray(np.asarray(fg_mask, dtype=bool)).reshape(-1)
    if fg.size != n:
        raise ValueError("mask and image sizes differ")
    return ctx.to_device(fg.view(np.uint8))


def _stats(pred, ref, mask_buf, thr=float("inf")):
    if pred.n != ref.n:
        raise ValueError("images must have the same number of voxels")
    return pred.ctx.masked_error_stats(pred.buf, pred.dtype, ref.buf, ref.dtype, mask_buf, pred.n,
                                       thr)


def _split_mae(out, n):
    n_fg = int(out[2])
    n_bg = n - n_fg
    fg_mae = float(out[0] / n_fg) if n_fg else 0.0
    bg_mae = float(out[1] / n_bg) if n_bg else 0.0
    return fg_mae, bg_mae


def foreground_background_mae(pred, ref, fg_mask):
    """(foreground MAE, background MAE) of ``pred`` against ``ref`` (reference metrics.py:306-330);
    a side with no voxels reports 0."""
    with DeviceImage(pred) as p, DeviceImage(ref, p.ctx) as r:
        m = _mask_buffer(p.ctx, fg_mask, p.n)
        try:
            return _split_mae(_stats(p, r, m), p.n)
        finally:
            m.free()


def _mip_error(out, pred, raw):
    """``float(abs(np.max(pred) - np.max(raw)))`` with numpy's scalar arithmetic of the callers'
    dtypes -- including the reference's wrap-around when both are uint16 and the prediction's
    maximum is the smaller one (metrics.py:349 subtracts two uint16 scalars)."""
    pm, rm = np.float64(out[4]), np.float64(out[5])
    if pred.orig_dtype.kind in "iu":
        pm = pred.orig_dtype.type(pm)
    if raw.orig_dtype.kind in "iu":
        rm = raw.orig_dtype.type(rm)
    if pred.orig_dtype == np.float32:
        pm = np.float32(pm)
    if raw.orig_dtype == np.float32:
        rm = np.float32(rm)
    with np.errstate(over="ignore"):
        return float(abs(pm - rm))


def mip_max_error(pred, raw):
    """|max(pred) - max(raw)| (reference metrics.py:333-349)."""
    with DeviceImage(pred) as p, DeviceImage(raw, p.ctx) as r:
        return _mip_error(_stats(p, r, None), p, r)


def _bright_threshold(raw, k):
    med = order_stats.median(raw.order_stats())
    mad = raw.median_abs_deviation(med) + 1e-6
    return med + k * 1.4826 * mad


def false_bright_rate(pred, raw, fg_mask, k=6.0):
    """Fraction of background voxels where ``pred`` exceeds median(raw) + k * 1.4826 * MAD(raw)
    (reference metrics.py:352-381)."""
    with DeviceImage(pred) as p, DeviceImage(raw, p.ctx) as r:
        m = _mask_buffer(p.ctx, fg_mask, p.n)
        try:
            out = _stats(p, r, m, float("inf"))
            n_bg = p.n - int(out[2])
            if not n_bg:
                return 0.0
            thr = _bright_threshold(r, k)
            out = _stats(p, r, m, float(thr))
            return float(out[3] / n_bg)
        finally:
            m.free()


def evaluate_example(pred, raw, target, fg_mask, pct=0.1):
    """The metric dictionary of one example, in counts (reference metrics.py:384-424): foreground
    fidelity against ``raw``, background cleanup against the BM4D ``target``, bright-tail
    percentile error / preservation, MIP maximum error and the false-bright rate."""
    with DeviceImage(pred) as p, DeviceImage(raw, p.ctx) as r, DeviceImage(target, p.ctx) as t:
        m = _mask_buffer(p.ctx, fg_mask, p.n)
        try:
            n_bg = p.n - int(_stats(p, r, m)[2])
            thr = _bright_threshold(r, 6.0) if n_bg else float("inf")
            vs_raw = _stats(p, r, m, float(thr))
            vs_target = _stats(p, t, m)
        finally:
            m.free()
        fg_mae, _ = _split_mae(vs_raw, p.n)
        _, bg_mae = _split_mae(vs_target, p.n)
        q = 100.0 - pct
        raw_top = float(order_stats.percentile(r.order_stats(), q))
        pred_top = float(order_stats.percentile(p.order_stats(), q))
        return {
            "fg_mae": fg_mae,
            "bg_mae": bg_mae,
            "top_pct_error": abs(pred_top - raw_top),
            "top_pct_preservation": pred_top / (raw_top + 1e-8),
            "mip_max_error": _mip_error(vs_raw, p, r),
            "false_bright_rate": float(vs_raw[3] / n_bg) if n_bg else 0.0,
        }


def checkpoint_score(metrics, cratio, weights=None):
    """Checkpoint-selection score, lower is better (reference metrics.py:427-450)."""
    w = DEFAULT_CHECKPOINT_WEIGHTS if weights is None else weights
    return (
        w.get("fg_mae", 0.0) * metrics["fg_mae"]
        + w.get("bg_mae", 0.0) * metrics["bg_mae"]
        + w.get("top_pct_error", 0.0) * metrics["top_pct_error"]
        - w.get("cratio", 0.0) * cratio
    )
