"""Order statistics from device histograms.

The reference takes every robust statistic with numpy on the host --
``np.percentile`` in ``estimate_offset`` (machine_learning/transforms.py:414-438),
``scripts/estimate_background_offsets.py:31-67`` and ``evaluate_example``
(machine_learning/metrics.py:413-415), ``np.median`` twice in ``false_bright_rate``
(metrics.py:375-377) -- which partitions a float copy of the whole volume.  Here the volume stays
in HBM: a HIP kernel produces its exact histogram (uint16 counts) or the digit histograms of an
order-preserving key (float data), and the few scalar steps that follow are done on the host
*with the same numpy scalar arithmetic numpy's quantile code uses*, so the results are the
reference's bit for bit:

* ``np.percentile`` of a float array divides ``q`` by ``dtype.type(100)`` -- for the float32 copy
  ``estimate_offset`` makes, the quantile, the virtual index ``(n - 1) * q`` and the
  interpolation weight are all float32;
* the interpolation is ``a + (b - a) * t``, replaced by ``b - (b - a) * (1 - t)`` when ``t >= 0.5``;
* a virtual index at or beyond ``n - 1`` selects the last element.
"""
import numpy as np


class OrderStats:
    """k-th smallest value (0-based) of a population described by sorted distinct ``values`` and
    their ``counts``."""

    def __init__(self, values, counts):
        counts = np.asarray(counts, dtype=np.int64)
        keep = counts > 0
        self.values = np.asarray(values)[keep]
        self.cum = np.cumsum(counts[keep])
        self.n = int(self.cum[-1]) if self.cum.size else 0

    def at(self, k):
        if k < 0:
            k += self.n
        return self.values[int(np.searchsorted(self.cum, k, side="right"))]


def from_u16_hist(hist, ignore_zeros=False, dtype=np.float32):
    """Order statistics of a uint16 volume from its 65536-bin histogram, as values of ``dtype``
    (the dtype the reference converts the sample to before calling numpy).  ``ignore_zeros``
    drops the zero bin unless nothing else is populated (transforms.py:433-436)."""
    counts = np.array(hist, dtype=np.int64)
    if ignore_zeros and counts[1:].sum() > 0:
        counts[0] = 0
    return OrderStats(np.arange(65536).astype(dtype), counts)


def percentile(stats, q, at=None):
    """``np.percentile(x, q)`` (method "linear") for the population behind ``stats``; the element
    dtype is ``stats.values.dtype`` unless a lookup ``at(k) -> numpy scalar`` is supplied."""
    n = stats.n
    at = at or stats.at
    if n == 0:
        raise ValueError("percentile of an empty sample")
    dt = np.dtype(type(at(0)))
    qq = np.asanyarray(np.true_divide(q, dt.type(100) if dt.kind == "f" else 100))
    if not (0 <= qq <= 1):
        raise ValueError("Percentiles must be in the range [0, 100]")
    virtual = np.asanyarray((n - 1) * qq)
    if virtual.dtype.kind in "iu":          # integer q on integer data: no interpolation
        return at(int(virtual))
    prev = np.floor(virtual)
    nxt = prev + 1
    if virtual >= n - 1:
        prev, nxt = -1, -1
    elif virtual < 0:
        prev, nxt = 0, 0
    prev_i, nxt_i = int(prev), int(nxt)
    gamma = np.asanyarray(virtual - np.intp(prev_i)).astype(virtual.dtype)
    a, b = np.asanyarray(at(prev_i)), np.asanyarray(at(nxt_i))
    diff = np.subtract(b, a)
    out = np.add(a, diff * gamma)
    if gamma >= 0.5:
        out = np.subtract(b, diff * (1 - gamma)).astype(out.dtype)
    return out[()]


def median(stats, at=None):
    """``np.median``: the middle element, or the mean of the two middle elements."""
    n = stats.n
    at = at or stats.at
    if n % 2:
        return np.float64(at(n // 2))
    return np.mean(np.array([at(n // 2 - 1), at(n // 2)], dtype=np.float64))


def median_abs_deviation(stats, center):
    """``np.median(np.abs(x - center))`` for float64 ``x`` (metrics.py:376)."""
    dev = np.abs(stats.values.astype(np.float64) - np.float64(center))
    order = np.argsort(dev, kind="stable")
    counts = np.diff(np.concatenate(([0], stats.cum)))[order]
    dev = dev[order]
    # merge equal deviations (v below and above the centre)
    uniq, inv = np.unique(dev, return_inverse=True)
    merged = np.bincount(inv, weights=counts).astype(np.int64)
    return median(OrderStats(uniq, merged))


# ---- data a 65536-bin histogram cannot hold: radix selection on the device ----------------------------
def _key_to_f64(key):
    bits = (key ^ (1 << 63)) if key >> 63 else (~key & 0xFFFFFFFFFFFFFFFF)
    return np.array([bits], dtype=np.uint64).view(np.float64)[0]


class DeviceOrderStats:
    """Exact order statistics of a device buffer (uint16 / float32 / float64 elements, widened to
    float64), or of the absolute deviations ``|x - center|``, by radix selection over the four
    16-bit digits of an order-preserving key: each digit histogram is one pass of
    ``exabm4d_key_histogram_dev`` restricted to the prefix chosen so far, and is cached.  Values
    are returned as ``dtype`` scalars (float32 data come back exactly as float32)."""

    def __init__(self, ctx, d_vol, elem_dtype, n, center=None, dtype=np.float64):
        self.ctx, self.d_vol, self.elem, self.n = ctx, d_vol, np.dtype(elem_dtype), int(n)
        self.center = center
        self.dtype = np.dtype(dtype)
        self._cum = {}

    def _digits(self, digit, prefix):
        key = (digit, prefix)
        if key not in self._cum:
            hist = self.ctx.key_histogram(self.d_vol, self.elem, self.n, digit, prefix, self.center)
            self._cum[key] = np.cumsum(hist.astype(np.int64))
        return self._cum[key]

    def at(self, k):
        if k < 0:
            k += self.n
        if not 0 <= k < self.n:
            raise IndexError(k)
        prefix = 0
        for digit in range(4):
            cum = self._digits(digit, prefix)
            d = int(np.searchsorted(cum, k, side="right"))
            k -= int(cum[d - 1]) if d else 0
            prefix = (prefix << 16) | d
        return self.dtype.type(_key_to_f64(prefix))

    def count_not_positive(self):
        """Elements <= 0: every negative key (top digit < 0x8000) plus the elements equal to +0.0
        (key 0x8000_0000_0000_0000)."""
        below = int(self._digits(0, 0)[0x7FFF])
        prefix = 0x8000
        for digit in (1, 2):
            if int(self._digits(digit, prefix)[0]) == 0:
                return below
            prefix <<= 16
        return below + int(self._digits(3, prefix)[0])


class Shifted:
    """The population of ``stats`` without its ``skip`` smallest elements."""

    def __init__(self, stats, skip):
        self.stats, self.skip, self.n = stats, skip, stats.n - skip

    def at(self, k):
        return self.stats.at(k + self.skip if k >= 0 else k)
