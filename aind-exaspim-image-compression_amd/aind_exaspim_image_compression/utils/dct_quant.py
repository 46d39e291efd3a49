"""Block-DCT transform quantiser of the encode half.

BASELINE.json config 5 names a "3D wavelet/DCT quantise -> entropy encode" step after the
denoiser.  The reference has none (it passes the denoised uint16 volume to Blosc-zstd or JPEG-XL,
``evaluate.py:40``, ``utils/img_util.py:401-441``), so the step is specified here:
non-overlapping 8^3 blocks, the orthonormal 3-D DCT of the BM4D transforms, uniform quantisation
``idx = int32(rint(c / q))``.  Everything runs on the GPU (``exabm4d_dctq_*_dev``); the rate is
estimated from the exact symbol histogram of the indices (``exabm4d_i32_symbol_histogram_dev``):
the order-0 entropy of an escape code whose alphabet is [-32767, 32767] plus an escape symbol that
costs 32 raw bits -- a bound for a memoryless coder, not an entropy coder.
"""
import numpy as np

from aind_exaspim_image_compression import _native


def _blocks(shape):
    return tuple(-(-int(n) // 8) for n in shape)


def quantise(vol, q, device=None):
    """uint16 volume -> int32 DCT indices ``[nbz, nby, nbx, 512]`` (coefficient raster (uz, uy, ux))."""
    vol = np.ascontiguousarray(vol, dtype=np.uint16)
    if vol.ndim != 3:
        raise ValueError("expected a 3-D uint16 volume")
    ctx = _native.context(device)
    nb = _blocks(vol.shape)
    d_vol = ctx.to_device(vol)
    d_idx = ctx.alloc(int(np.prod(nb)) * 512 * 4)
    try:
        ctx.dctq_forward(d_vol, vol.shape, q, d_idx)
        ctx.sync()
        return d_idx.download(nb + (512,), np.int32)
    finally:
        d_vol.free()
        d_idx.free()


def reconstruct(idx, shape, q, device=None):
    """int32 indices -> uint16 volume of ``shape`` (dequantise, inverse DCT, clamp, round)."""
    shape = tuple(int(s) for s in shape)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    if idx.shape != _blocks(shape) + (512,):
        raise ValueError("index array does not match the volume shape")
    ctx = _native.context(device)
    d_idx = ctx.to_device(idx)
    d_vol = ctx.alloc(int(np.prod(shape)) * 2)
    try:
        ctx.dctq_inverse(d_idx, shape, q, d_vol)
        ctx.sync()
        return d_vol.download(shape, np.uint16)
    finally:
        d_idx.free()
        d_vol.free()


def _bits(hist, n_idx, nvox):
    """Order-0 bits per voxel of the escape code described by the symbol histogram."""
    hist = np.asarray(hist, dtype=np.float64)
    p = hist[hist > 0] / n_idx
    return float((-(p * np.log2(p)).sum() * n_idx + 32.0 * hist[0]) / nvox)


def entropy_bits_per_voxel(idx, nvox, device=None):
    """Order-0 rate estimate of the index stream in bits per volume voxel (device histogram)."""
    idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1)
    ctx = _native.context(device)
    buf = ctx.to_device(idx)
    try:
        hist = ctx.i32_symbol_histogram(buf, idx.size)
    finally:
        buf.free()
    return _bits(hist, idx.size, nvox)


def rate_distortion(vol, q, device=None):
    """One point of the R-D sweep: quantise, reconstruct, rate proxy, mean and maximum absolute
    error -- volume, indices and reconstruction stay in HBM between the kernels."""
    vol = np.ascontiguousarray(vol, dtype=np.uint16)
    ctx = _native.context(device)
    nb = _blocks(vol.shape)
    n_idx = int(np.prod(nb)) * 512
    d_vol = ctx.to_device(vol)
    d_idx = ctx.alloc(n_idx * 4)
    d_rec = ctx.alloc(vol.nbytes)
    try:
        ctx.dctq_forward(d_vol, vol.shape, q, d_idx)
        ctx.dctq_inverse(d_idx, vol.shape, q, d_rec)
        hist = ctx.i32_symbol_histogram(d_idx, n_idx)
        err = ctx.masked_error_stats(d_rec, np.uint16, d_vol, np.uint16, None, vol.size)
    finally:
        d_vol.free()
        d_idx.free()
        d_rec.free()
    return {"q": float(q), "bits_per_voxel": _bits(hist, n_idx, vol.size),
            "mae": float(err[1] / vol.size), "lmax": float(err[6])}
