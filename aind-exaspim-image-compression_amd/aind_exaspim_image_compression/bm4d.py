"""BM4D on MI355X behind the call signature the reference uses.

The reference imports the third-party wheel (``from bm4d import bm4d``) and calls it as the
opaque two-argument function ``bm4d(raw, sigma)`` (reference machine_learning/data_handling.py:12,
:332, :926; evaluate.py:11, :202).  ``bm4d`` below accepts exactly that call; the work is done by
the HIP kernels of ``libexabm4d.so`` with the profile of BASELINE.json (8^3 blocks, step 4, 11^3
search window, groups of <= 16, 3-D DCT + Haar, hard-threshold stage then Wiener stage --
DESIGN.md section 3).  There is no CPU fallback.

Harness-level helpers mirror the reference's two call patterns (SURVEY.md section 8 row a-J):
``denoise_patches`` = scripts/precompute.py teacher generation (batch of fp32 patches -> clipped
teachers) and ``denoise_volume`` = uint16 volume in, uint16 volume out.
"""
from dataclasses import dataclass, asdict

import numpy as np

from aind_exaspim_image_compression import _native


@dataclass(frozen=True)
class BM4DProfile:
    """Algorithm parameters (``exabm4d_params``).  Only the defaults' block / step / search /
    max_group values are implemented by the kernels; other values raise ``ValueError``."""

    block: int = 8
    step: int = 4
    search: int = 11
    max_group: int = 16
    lambda_ht: float = 2.7
    c_match_ht: float = 3.0
    c_match_wie: float = 0.6
    kaiser_beta: float = 2.0

    def native(self):
        return _native.default_params(**asdict(self))


_MAX_VOXELS_PER_CALL = 1 << 30     # ~21 GB of device scratch for the two-stage pipeline


def _denoise_batched(ctx, arr, sigma, params, stages, clip):
    """One device call per sub-batch of volumes (the reference's cache build runs 30 000
    patches, scripts/precompute.py:278-319; they do not have to fit the GPU at once)."""
    if arr.ndim == 3 or arr.shape[0] * arr[0].size <= _MAX_VOXELS_PER_CALL:
        return ctx.denoise_f32_host(arr, sigma, params=params, stages=stages, clip=clip)
    per = max(1, _MAX_VOXELS_PER_CALL // arr[0].size)
    out = np.empty(arr.shape, dtype=np.float32)
    for i in range(0, arr.shape[0], per):
        out[i:i + per] = ctx.denoise_f32_host(arr[i:i + per], sigma, params=params,
                                              stages=stages, clip=clip)
    return out


def _stages(stage_arg):
    if stage_arg in (None, "all", "ALL_STAGES", 2):
        return 2
    if stage_arg in ("ht", "hard_thresholding", "HARD_THRESHOLDING", 1):
        return 1
    raise ValueError(f"unknown stage_arg: {stage_arg!r}")


def bm4d(z, sigma_psd, profile=None, stage_arg=None, device=None):
    """Denoise a 3-D volume (or a 4-D batch of volumes) with two-stage BM4D.

    ``z``: array of counts, any real dtype (computed in float32, like the reference's patches:
    data_handling.py:353); ``sigma_psd``: noise standard deviation in the same units.  Returns a
    new float32 array of the same shape, NOT clipped (the reference clips at its call site,
    data_handling.py:333)."""
    arr = np.asarray(z)
    if arr.ndim not in (3, 4):
        raise ValueError("bm4d expects a 3-D volume or a 4-D batch of volumes")
    sigma = float(np.asarray(sigma_psd).reshape(-1)[0])
    prof = profile or BM4DProfile()
    ctx = _native.context(device)
    return _denoise_batched(ctx, arr.astype(np.float32, copy=False), sigma, prof.native(),
                            _stages(stage_arg), None)


def denoise_patches(raw, sigma, max_count=65535.0, profile=None, device=None):
    """``teacher = np.clip(bm4d(raw, sigma), 0, max_count)`` for a batch ``raw[N, Z, Y, X]`` of
    offset-subtracted float32 patches, in one device call (reference call pattern:
    data_handling.py:331-333 inside scripts/precompute.py:215-228)."""
    raw = np.asarray(raw, dtype=np.float32)
    if raw.ndim == 3:
        raw = raw[None]
    prof = profile or BM4DProfile()
    ctx = _native.context(device)
    return _denoise_batched(ctx, raw, float(sigma), prof.native(), 2, (0.0, float(max_count)))


def denoise_volume(vol_u16, sigma, offset=0.0, profile=None, stages=2, device=None):
    """uint16 volume -> uint16 volume: ``(float)v - offset`` -> BM4D -> ``+ offset`` -> clip to
    [0, 65535] -> rint -> uint16, entirely on the device (read_counts + bm4d + clip + the
    rint/uint16 cast of IntensityTransform.inverse)."""
    vol = np.ascontiguousarray(vol_u16, dtype=np.uint16)
    if vol.ndim != 3:
        raise ValueError("denoise_volume expects a 3-D uint16 volume")
    prof = profile or BM4DProfile()
    ctx = _native.context(device)
    d_in = ctx.to_device(vol)
    d_out = ctx.alloc(vol.nbytes)
    try:
        ctx.denoise_u16(d_in, d_out, vol.shape, float(sigma), float(offset), params=prof.native(),
                        stages=int(stages))
        ctx.sync()
        return d_out.download(vol.shape, np.uint16)
    finally:
        d_in.free()
        d_out.free()


def denoise_chunked(vol_u16, sigma, offset=0.0, chunk=256, halo=8, profile=None, stages=2,
                    device=None):
    """Chunk-local mode of BASELINE.json config 4: the volume is tiled by ``chunk``^3 cores, every
    core is read with ``halo`` voxels on each side -- the read window is CUT where the volume ends,
    nothing is padded or replicated at the faces (SURVEY.md appendix A item 11's
    "edge clamping" is read as clamping the window, and the oracle processes the identical
    truncated arrays) -- and denoised in isolation -- independent units, like the reference's one-``bm4d``-call-per-patch
    pool (scripts/precompute.py:215-228) -- and only the cores are written.  One batched device
    call (``exabm4d_denoise_chunked_u16_dev``); uint16 in, uint16 out."""
    vol = np.ascontiguousarray(vol_u16, dtype=np.uint16)
    if vol.ndim != 3:
        raise ValueError("denoise_chunked expects a 3-D uint16 volume")
    prof = profile or BM4DProfile()
    ctx = _native.context(device)
    d_in = ctx.to_device(vol)
    d_out = ctx.alloc(vol.nbytes)
    try:
        ctx.denoise_chunked_u16(d_in, d_out, vol.shape, float(sigma), float(offset),
                                chunk=int(chunk), halo=int(halo), params=prof.native(),
                                stages=int(stages))
        ctx.sync()
        return d_out.download(vol.shape, np.uint16)
    finally:
        d_in.free()
        d_out.free()


def denoise_chunked_streamed(vol_u16, sigma, offset=0.0, chunk=256, halo=8, profile=None, stages=2,
                             device=None, out=None):
    """``denoise_chunked`` for volumes that should not (or cannot) sit on the device whole -- the
    reference's production harness denoises whole images (scripts/evaluate_bm4dnet.py:51-181), and
    BASELINE config 4's tile is 64 GiB.  ``vol_u16`` is a C-contiguous uint16 array or ``np.memmap``;
    layers of chunks travel up, are denoised chunk by chunk in isolation and travel down while the
    next layer is in the kernels (``exabm4d_denoise_chunked_u16_host``).  ``out`` may name the
    destination (another array / writeable memmap of the same shape); the result equals
    ``denoise_chunked`` on the whole volume."""
    vol = vol_u16 if isinstance(vol_u16, np.ndarray) and vol_u16.dtype == np.uint16 and \
        vol_u16.flags.c_contiguous else np.ascontiguousarray(vol_u16, dtype=np.uint16)
    if vol.ndim != 3:
        raise ValueError("denoise_chunked_streamed expects a 3-D uint16 volume")
    if out is None:
        out = np.empty(vol.shape, dtype=np.uint16)
    _native.check_host_volume_pair(vol, out)
    prof = profile or BM4DProfile()
    ctx = _native.context(device)
    ctx.denoise_chunked_u16_host(vol, out, float(sigma), float(offset), chunk=int(chunk), halo=int(halo),
                                 params=prof.native(), stages=int(stages))
    return out
