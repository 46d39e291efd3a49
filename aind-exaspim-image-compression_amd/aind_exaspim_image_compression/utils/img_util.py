"""Compression-ratio operators (SURVEY.md section 8 "next" row f-1).

``compute_cratio`` keeps the reference's signature and semantics (reference
``utils/img_util.py:401-441``): C-order chunks of ``patch_shape``, ``codec.encode(chunk)``,
ratio rounded to two decimals.  The codec is whatever object the caller passes -- the reference
passes ``numcodecs.blosc.Blosc(cname="zstd", clevel=5|6, shuffle=SHUFFLE)`` (``evaluate.py:40``,
``scripts/evaluate_bm4dnet.py:140``), which is third-party and not part of this repo.

``utils.chunk_codec.ExacCodec`` is a codec object of that shape whose arithmetic runs on the
MI355X (EXAC v2: predictive, context-modelled rANS; ``version=1`` / ``ShuffleRansCodec``: byte
shuffle + order-0 rANS per byte plane); given it, ``compute_cratio`` codes all chunks in one
batched device call.

``shuffled_entropy_cratio`` is the MI355X-side rate floor of that codec: a HIP kernel builds, per 64^3 chunk,
the histograms of the two byte planes Blosc's SHUFFLE filter produces; the zeroth-order entropy
of those planes bounds what an order-0 entropy coder behind the shuffle can reach.  It is a proxy
for rate-distortion sweeps that keeps the volume in HBM, NOT the Blosc/zstd byte count (zstd also
exploits repeats); parity of the histograms with numpy is exact and tested.
"""
import numpy as np

from aind_exaspim_image_compression import _native


def compute_cratio(img, codec, patch_shape=(64, 64, 64)):
    """Chunked compression ratio = total raw bytes / total ``codec.encode`` bytes."""
    img = np.asarray(img)
    if img.ndim == 5:
        img = img[0, 0]
    img = np.ascontiguousarray(img, dtype=np.uint16)
    if img.ndim == 3 and hasattr(codec, "chunk_sizes"):
        # a device codec (utils/chunk_codec.ExacCodec): all chunks in one batched call;
        # the sizes are exactly len(codec.encode(chunk)) of the loop below
        return round(img.nbytes / int(codec.chunk_sizes(img, patch_shape).sum(dtype=np.uint64)), 2)
    raw = 0
    packed = 0
    grids = [range(0, s, c) for s, c in zip(img.shape, patch_shape)]
    for z0 in grids[0]:
        for y0 in grids[1]:
            for x0 in (grids[2] if len(grids) > 2 else [0]):
                chunk = np.ascontiguousarray(img[z0:z0 + patch_shape[0], y0:y0 + patch_shape[1],
                                                 x0:x0 + patch_shape[2]])
                packed += len(codec.encode(chunk))
                raw += chunk.nbytes
    return round(raw / packed, 2)


def chunk_byte_histograms(img, patch_shape=(64, 64, 64), device=None):
    """[nchunks, 2, 256] uint32 histograms of the low / high byte planes of every chunk (GPU)."""
    vol = np.ascontiguousarray(img, dtype=np.uint16)
    if vol.ndim != 3:
        raise ValueError("expected a 3-D uint16 volume")
    ctx = _native.context(device)
    n = int(np.prod([-(-s // c) for s, c in zip(vol.shape, patch_shape)]))
    d_vol = ctx.to_device(vol)
    d_hist = ctx.alloc(n * 512 * 4)
    try:
        ctx.chunk_byte_histograms(d_vol, vol.shape, patch_shape, d_hist)
        ctx.sync()
        return d_hist.download((n, 2, 256), np.uint32)
    finally:
        d_vol.free()
        d_hist.free()


def entropy_bytes(hist):
    """Zeroth-order entropy bound in bytes of the byte planes described by ``hist[..., 256]``."""
    h = np.asarray(hist, dtype=np.float64)
    n = h.sum(axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        bits = -np.where(h > 0, h * np.log2(h / n), 0.0).sum(axis=-1)
    return bits / 8.0


def shuffled_entropy_cratio(img, patch_shape=(64, 64, 64), device=None):
    """Raw bytes / order-0 entropy bound of the byte-shuffled chunks, rounded like compute_cratio."""
    hist = chunk_byte_histograms(img, patch_shape, device)
    raw = float(hist[:, 0, :].sum()) * 2.0
    return round(raw / max(float(entropy_bytes(hist).sum()), 1.0), 2)


# ---- quality metrics on device ----------------------------------------
def _device_pair(img1, img2):
    """Upload two images with one common element type: uint16 when both are, float64 otherwise
    (the reference widens everything to float64: img_util.py:981-982)."""
    a, b = np.asarray(img1), np.asarray(img2)
    if a.shape != b.shape:
        raise ValueError("Input images must have the same dimensions")
    if not (a.dtype == np.uint16 and b.dtype == np.uint16):
        a, b = a.astype(np.float64), b.astype(np.float64)
    ctx = _native.context()
    return ctx, a, ctx.to_device(np.ascontiguousarray(a).reshape(-1)), \
        ctx.to_device(np.ascontiguousarray(b).reshape(-1))


def ssim3D(img1, img2, data_range=None, window_size=16):
    """Structural similarity of two 3-D images with a cubic uniform window (reference
    ``utils/img_util.py:953-1003``): local moments by ``scipy.ndimage.uniform_filter`` semantics
    ("reflect" boundary), ``C1 = (0.01 L)^2``, ``C2 = (0.03 L)^2``, mean of
    ``num / (max(den, 1e-8) + 1e-6)``.

    The moments and the SSIM map are formed by one fp64 HIP kernel that marches running box sums
    along z (``exabm4d_ssim3d_dev``); for uint16 input and the default power-of-two window the
    local moments are exact, and the result differs from the reference only by the summation order
    of the final mean (relative 1e-15)."""
    ctx, a, d_a, d_b = _device_pair(img1, img2)
    try:
        if a.ndim != 3:
            raise ValueError("ssim3D expects 3-D images")
        n = int(a.size)
        if data_range is None:
            lo1, hi1 = ctx.minmax(d_a, a.dtype, n)
            lo2, hi2 = ctx.minmax(d_b, a.dtype, n)
            data_range = max(hi1 - lo1, hi2 - lo2)
        c1 = (0.01 * data_range) ** 2
        c2 = (0.03 * data_range) ** 2
        total = ctx.ssim3d_sum(d_a, d_b, a.dtype, a.shape, window_size, c1, c2)
        return np.float64(total) / n
    finally:
        d_a.free()
        d_b.free()


def _abs_error_stats(img1, img2):
    ctx, a, d_a, d_b = _device_pair(img1, img2)
    try:
        n = int(a.size)
        return ctx.masked_error_stats(d_a, a.dtype, d_b, a.dtype, None, n), n
    finally:
        d_a.free()
        d_b.free()


def compute_mae(img1, img2):
    """Mean absolute error (reference ``utils/img_util.py`` compute_mae)."""
    out, n = _abs_error_stats(img1, img2)
    return float(out[1] / n)


def compute_lmax(img1, img2):
    """Maximum absolute error (reference ``utils/img_util.py`` compute_lmax)."""
    out, _ = _abs_error_stats(img1, img2)
    return float(out[6])
