"""Count <-> normalised-intensity transforms, evaluated by fused HIP kernels on MI355X.

Drop-in for the reference module ``machine_learning/transforms.py`` (same class names,
constructor arguments, attributes, cfg-dict schema and error behaviour; reference lines are cited
per item).  The arithmetic of ``forward`` / ``inverse`` / ``inverse_float`` runs in
``libexabm4d.so`` (``csrc/elementwise_kernels.hip``) -- numpy arrays are copied to the GPU, the
kernel runs, the result is copied back; there is no CPU fallback.  Code that already holds device
buffers uses ``forward_device`` / ``inverse_device`` and never leaves HBM
(``inference.predict`` does).

Only construction-time scalars (e.g. the normalisation constant) are computed on the host,
exactly where the reference computes them on the host; ``estimate_offset`` reduces its sample to
a histogram on the GPU and finishes the percentile on the host.
"""
import numpy as np

from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.utils import order_stats

_F32 = np.float32
_KIND_IDS = {"asinh": 0, "anscombe": 1, "linear": 2}


def _run_elementwise(tf, x, direction):
    """numpy in -> HIP kernel -> numpy out, preserving the input's shape."""
    ctx = _native.context()
    x = np.asarray(x)
    shape = x.shape
    if direction == "forward":
        src_u16 = x.dtype == np.uint16
        flat = np.ascontiguousarray(x if src_u16 else x.astype(_F32)).reshape(-1)
    else:
        src_u16 = False
        flat = np.ascontiguousarray(x, dtype=_F32).reshape(-1)
    n = flat.size
    if n == 0:
        return np.empty(shape, dtype=np.uint16 if direction == "inverse" else _F32)
    d_in = ctx.to_device(flat)
    spec = tf.native_struct()
    if direction == "forward":
        d_out = ctx.alloc(4 * n)
        ctx.transform_forward(spec, d_in, d_out, n, src_u16)
        out = d_out.download((n,), _F32)
    elif direction == "inverse":
        d_out = ctx.alloc(2 * n)
        ctx.transform_inverse(spec, d_in, d_out, n, quantise=True)
        out = d_out.download((n,), np.uint16)
    else:
        d_out = ctx.alloc(4 * n)
        ctx.transform_inverse(spec, d_in, d_out, n, quantise=False)
        out = d_out.download((n,), _F32)
    d_in.free()
    d_out.free()
    return out.reshape(shape)


class IntensityTransform:
    """Abstract base (reference transforms.py:23-62)."""

    def forward(self, x):
        """Raw counts -> normalised domain (float32)."""
        raise NotImplementedError

    def inverse(self, y):
        """Normalised domain -> raw uint16 counts, clipped to [0, max_count]."""
        raise NotImplementedError

    def inverse_float(self, y):
        """Normalised domain -> unclipped floating-point counts."""
        raise NotImplementedError


class _DeviceTransform(IntensityTransform):
    """Shared GPU plumbing of the concrete transforms."""

    def _fill(self, spec):
        raise NotImplementedError

    def native_struct(self):
        """The ``exabm4d_transform`` descriptor of this object (include/exabm4d.h)."""
        import ctypes
        spec = _native.Transform()
        spec.size = ctypes.sizeof(_native.Transform)
        spec.gain = 1.0  # keeps 2/gain finite for the kinds that ignore it
        self._fill(spec)
        return spec

    def forward(self, x):
        return _run_elementwise(self, x, "forward")

    def inverse(self, y):
        return _run_elementwise(self, y, "inverse")

    def inverse_float(self, y):
        return _run_elementwise(self, y, "inverse_float")

    # -- device-resident variants (no host round trip) ---------------------------------------
    def forward_device(self, ctx, src, dst, n, src_is_u16):
        ctx.transform_forward(self.native_struct(), src, dst, n, src_is_u16)

    def inverse_device(self, ctx, src, dst, n, quantise=True):
        ctx.transform_inverse(self.native_struct(), src, dst, n, quantise=quantise)


class AsinhTransform(_DeviceTransform):
    """HDR-style asinh transform (reference transforms.py:65-152).

    ``forward(x) = asinh((x - offset) / scale) / asinh((max_count - offset) / scale)``.
    """

    def __init__(self, offset=0.0, scale=32.0, max_count=65535.0):
        self.offset = float(offset)
        self.scale = float(scale)
        self.max_count = float(max_count)
        self._norm = float(np.arcsinh((self.max_count - self.offset) / self.scale))

    def _fill(self, spec):
        spec.kind = _KIND_IDS["asinh"]
        spec.max_count = self.max_count
        spec.offset = self.offset
        spec.scale = self.scale
        spec.norm = self._norm


def _gat_scalar_f32(gain, read_noise, offset, x):
    """The generalised Anscombe transform of one value with numpy's fp32 rounding points
    (reference transforms.py:223-242 applied to a 0-d float32 array)."""
    arg = _F32(gain) * (_F32(x) - _F32(offset))
    arg = arg + _F32((3.0 / 8.0) * gain ** 2)
    arg = arg + _F32(read_noise ** 2)
    return _F32(2.0 / gain) * np.sqrt(np.maximum(arg, _F32(0.0)))


class AnscombeTransform(_DeviceTransform):
    """Generalised Anscombe variance-stabilising transform (reference transforms.py:155-285)."""

    def __init__(self, gain=1.0, read_noise=0.0, offset=0.0, max_count=65535.0,
                 unbiased_inverse=True):
        self.gain = float(gain)
        self.read_noise = float(read_noise)
        self.offset = float(offset)
        self.max_count = float(max_count)
        self.unbiased_inverse = bool(unbiased_inverse)
        self._c_inv = 1.0 / 8.0 if unbiased_inverse else 3.0 / 8.0
        self._norm = float(_gat_scalar_f32(self.gain, self.read_noise, self.offset,
                                           self.max_count))

    def _fill(self, spec, norm=None):
        spec.kind = _KIND_IDS["anscombe"]
        spec.max_count = self.max_count
        spec.offset = self.offset
        spec.gain = self.gain
        spec.read_noise = self.read_noise
        spec.c_inv = self._c_inv
        spec.norm = self._norm if norm is None else norm

    def _gat(self, x):
        """Unnormalised GAT (reference transforms.py:223-242), on the GPU."""
        return _run_elementwise(_UnnormalisedGat(self), x, "forward")


class _UnnormalisedGat(_DeviceTransform):
    """An AnscombeTransform evaluated with a normalisation constant of exactly 1."""

    def __init__(self, parent):
        self.parent = parent

    def _fill(self, spec):
        self.parent._fill(spec, norm=1.0)


class LinearClipTransform(_DeviceTransform):
    """Linear normalisation with a hard brightness clip (reference transforms.py:288-371)."""

    def __init__(self, mn=0.0, mx=1000.0, clip=8.0, max_count=65535.0):
        self.mn = float(mn)
        self.mx = float(mx)
        self.clip = float(clip)
        self.max_count = float(max_count)

    def _fill(self, spec):
        spec.kind = _KIND_IDS["linear"]
        spec.max_count = self.max_count
        spec.mn = self.mn
        spec.mx = self.mx
        spec.clip = self.clip


class OffsetTransform(_DeviceTransform):
    """``forward(x) = base.forward(x - offset)``, ``inverse(y) = base.inverse_float(y) + offset``
    (reference transforms.py:374-411).  The base transform's normalisation is left untouched."""

    def __init__(self, base_transform, offset=0.0):
        self.base_transform = base_transform
        self.offset = float(offset)
        self.max_count = float(base_transform.max_count)

    def __getattr__(self, name):
        # non-offset parameters (scale, gain, ...) come from the base (transforms.py:394-396)
        if name == "base_transform":
            raise AttributeError(name)
        return getattr(self.base_transform, name)

    def _fill(self, spec):
        self.base_transform._fill(spec)
        spec.wrapped = 1
        spec.wrap_offset = self.offset
        spec.max_count = self.max_count


def estimate_offset(sample, percentile=1.0, ignore_zeros=True):
    """Robust background / black-point estimate in counts (reference transforms.py:414-438):
    ``np.percentile`` of the float32 sample, zeros (non-positive values) excluded unless nothing
    else is there.

    The sample is reduced on the GPU: a uint16 sample to its exact
    65536-bin histogram, a float sample to the digit histograms of a radix selection over its float32 values; the
    percentile is then formed on the host with numpy's own float32 scalar steps
    (``utils/order_stats.py``), so the value is the reference's bit for bit."""
    sample = np.asarray(sample)
    if sample.size == 0:
        raise ValueError("estimate_offset needs a non-empty sample")
    ctx = _native.context()
    if sample.dtype == np.uint16:
        flat = np.ascontiguousarray(sample).reshape(-1)
        buf = ctx.to_device(flat)
        try:
            return estimate_offset_device(ctx, buf, flat.size, percentile, ignore_zeros)
        finally:
            buf.free()
    flat = np.ascontiguousarray(sample, dtype=_F32).reshape(-1)
    buf = ctx.to_device(flat)
    try:
        stats = order_stats.DeviceOrderStats(ctx, buf, _F32, flat.size, dtype=_F32)
        if ignore_zeros:
            skip = stats.count_not_positive()
            if skip < stats.n:
                stats = order_stats.Shifted(stats, skip)
        return float(order_stats.percentile(stats, percentile))
    finally:
        buf.free()


def estimate_offset_device(ctx, d_u16, n, percentile=1.0, ignore_zeros=True):
    """``estimate_offset`` of ``n`` uint16 voxels already resident in HBM (device pointer,
    ``DeviceBuffer`` or torch tensor)."""
    hist = ctx.u16_histogram(d_u16, int(n))
    stats = order_stats.from_u16_hist(hist, ignore_zeros=ignore_zeros, dtype=_F32)
    return float(order_stats.percentile(stats, percentile))


def background_offset_statistics(sample, percentile=0.1):
    """The per-brain record of ``scripts/estimate_background_offsets.py:31-67`` for a uint16
    volume: offset over the non-zero voxels, offset over all voxels, median of the non-zero voxels
    and the fraction of zero voxels, from one device histogram."""
    flat = np.ascontiguousarray(sample, dtype=np.uint16).reshape(-1)
    ctx = _native.context()
    buf = ctx.to_device(flat)
    try:
        hist = ctx.u16_histogram(buf, flat.size)
    finally:
        buf.free()
    n = int(hist.sum())
    nonzero = n - int(hist[0])
    every = order_stats.from_u16_hist(hist, dtype=np.uint16)
    if nonzero:
        counts = hist.copy()
        counts[0] = 0
        pos = order_stats.from_u16_hist(counts, dtype=np.uint16)
        offset = float(order_stats.percentile(pos, percentile))
        med = float(order_stats.median(pos))
    else:
        offset = med = float("nan")
    return {
        "offset": offset,
        "offset_all_voxels": float(order_stats.percentile(every, percentile)),
        "median": med,
        "zero_fraction": 1.0 - nonzero / n,
    }


_BUILDERS = {
    "asinh": AsinhTransform,
    "anscombe": AnscombeTransform,
    "linear": LinearClipTransform,
}


def build_transform(cfg):
    """Build a transform from ``{"kind", "params"[, "base"]}`` and stamp the frozen cfg on it as
    ``.cfg`` (reference transforms.py:441-481).  Unknown kinds raise ``ValueError``."""
    kind = cfg["kind"]
    params = cfg.get("params", {})
    if kind in _BUILDERS:
        transform = _BUILDERS[kind](**params)
    elif kind == "offset":
        transform = OffsetTransform(build_transform(cfg["base"]), **params)
    else:
        raise ValueError(f"Unknown transform kind: {kind}")
    transform.cfg = {**cfg, "params": dict(params)}
    return transform


def calibrate_transform(cfg, sample):
    """Freeze the data-driven black-point into a NEW cfg (reference transforms.py:484-513)."""
    out = {**cfg, "params": dict(cfg.get("params", {}))}
    calib = out.get("calibrate", {})
    if calib.get("offset", False):
        out["params"]["offset"] = estimate_offset(
            sample, percentile=calib.get("offset_percentile", 1.0))
    return out


def with_offset(transform, offset):
    """Compose a raw-count background offset around a trained transform (reference
    transforms.py:516-562): linear transforms get both bounds shifted, everything else is
    wrapped in an ``OffsetTransform``; an existing wrapper is replaced, not nested."""
    if isinstance(transform, OffsetTransform):
        transform = transform.base_transform
    cfg = getattr(transform, "cfg", None)
    if cfg is None:
        raise ValueError("transform has no cfg; construct it via build_transform")
    offset = float(offset)
    if cfg["kind"] == "linear":
        params = dict(cfg.get("params", {}))
        params["mn"] = float(transform.mn) + offset
        params["mx"] = float(transform.mx) + offset
        return build_transform({**cfg, "params": params})
    return build_transform({"kind": "offset", "base": cfg, "params": {"offset": offset}})
