"""Block-DCT transform quantiser of the encode half.

BASELINE.json config 5 names a "3D wavelet/DCT quantise -> entropy encode" step after the
denoiser.  The reference has none (it passes the denoised uint16 volume to Blosc-zstd or JPEG-XL,
``evaluate.py:40``, ``utils/img_util.py:401-441``), so the step is specified here:
non-overlapping 8^3 blocks, the orthonormal 3-D DCT of the BM4D transforms, uniform quantisation
``idx = int32(rint(c / q))``.  Everything runs on the GPU (``exabm4d_dctq_*_dev``).  The rate of a
point is the size of the real byte streams the chunk coder makes of the indices
(``exabm4d_codec_encode_dev``, EXAC v2, chunks of 512 blocks x 8 x 64 coefficients -- what bench.py
times); the order-0 entropy of an escape code over the index histogram
(``exabm4d_i32_symbol_histogram_dev``) is kept next to it as the memoryless bound it beats.
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


INDEX_CHUNK = (512, 8, 64)      # 2^18 indices per coded chunk: 512 blocks x (uz) x (uy, ux)


def rate_distortion_device(ctx, d_vol, shape, q, d_idx=None, d_rec=None):
    """One point of the R-D sweep on a uint16 volume that already lies in HBM: quantise, code the
    indices (real bytes), reconstruct, mean / maximum absolute error and sum of squared errors are
    not needed on the host -- everything stays on the device.  ``d_idx`` / ``d_rec`` may be
    caller-owned scratch (4 / 2 bytes per voxel of the block-padded volume)."""
    shape = tuple(int(s) for s in shape)
    nb = _blocks(shape)
    nblk = int(np.prod(nb))
    n_idx, nvox = nblk * 512, int(np.prod(shape))
    own = []
    if d_idx is None:
        d_idx = ctx.alloc(n_idx * 4)
        own.append(d_idx)
    if d_rec is None:
        d_rec = ctx.alloc(nvox * 2)
        own.append(d_rec)
    try:
        ctx.dctq_forward(d_vol, shape, q, d_idx)
        coded, _ = ctx.codec_encode(d_idx, 4, (nblk, 8, 64), INDEX_CHUNK)
        ctx.dctq_inverse(d_idx, shape, q, d_rec)
        hist = ctx.i32_symbol_histogram(d_idx, n_idx)
        err = ctx.masked_error_stats(d_rec, np.uint16, d_vol, np.uint16, None, nvox)
    finally:
        for b in own:
            b.free()
    return {"q": float(q), "coded_bytes": int(coded), "bits_per_voxel": 8.0 * coded / nvox,
            "order0_bits_per_voxel": _bits(hist, n_idx, nvox),
            "mae": float(err[1] / nvox), "lmax": float(err[6])}


def rate_distortion(vol, q, device=None):
    """``rate_distortion_device`` for a host volume."""
    vol = np.ascontiguousarray(vol, dtype=np.uint16)
    ctx = _native.context(device)
    d_vol = ctx.to_device(vol)
    try:
        return rate_distortion_device(ctx, d_vol, vol.shape, q)
    finally:
        d_vol.free()
