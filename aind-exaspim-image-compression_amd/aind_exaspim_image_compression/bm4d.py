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

Placement (round 4).  The reference's pattern is a pool of forked workers, one ``bm4d(raw, sigma)`` per 64^3
patch (scripts/precompute.py:215-228).  Three ways to run it on an 8-GPU node, slowest first:
  * unchanged: every worker opens a context on ``_native.default_device()`` = its pool index mod the device
    count (``EXABM4D_DEVICE`` overrides) -- the pool spreads over the GPUs instead of piling onto device 0;
  * ``EXABM4D_BROKER=1`` (or ``broker.enable()`` in the parent): the workers never touch a GPU; one owner
    process per device coalesces their single patches into batched calls (``broker.py``);
  * ``denoise_patches(raw, sigma, devices="all")``: the whole batch split over all GPUs in one call.
All three give the same teachers, bit for bit (every patch carries its own fixed-point unit).
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


def _denoise_batched(device, arr, sigma, params, stages, clip):
    """One device call per sub-batch of volumes (the reference's cache build runs 30 000
    patches, scripts/precompute.py:278-319; they do not have to fit the GPU at once).  With the broker
    enabled (and no explicit device) the calls go to the device's owner process instead of a context of
    this process."""
    from aind_exaspim_image_compression import broker
    if device is None and broker.enabled():
        def run(a):
            return broker.denoise(a, sigma, params, stages, clip)
    else:
        ctx = _native.context(device)

        def run(a):
            return ctx.denoise_f32_host(a, sigma, params=params, stages=stages, clip=clip)
    if arr.ndim == 3 or arr.shape[0] * arr[0].size <= _MAX_VOXELS_PER_CALL:
        return run(arr)
    per = max(1, _MAX_VOXELS_PER_CALL // arr[0].size)
    out = np.empty(arr.shape, dtype=np.float32)
    for i in range(0, arr.shape[0], per):
        out[i:i + per] = run(arr[i:i + per])
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
    return _denoise_batched(device, arr.astype(np.float32, copy=False), sigma, prof.native(),
                            _stages(stage_arg), None)


def split_batch(n, parts):
    """[start, stop) of ``parts`` near-equal consecutive shares of ``n`` items (empty shares dropped)."""
    parts = max(1, min(int(parts), int(n))) if n > 0 else 1
    edges = [(n * i) // parts for i in range(parts + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _device_share(args):
    """Child process of ``denoise_patches(..., devices=...)``: one device, one share of the batch, data in
    the parent's shared-memory segments."""
    from multiprocessing import shared_memory
    name_in, name_out, shape, a, b, sigma, max_count, prof, device = args
    seg_in, seg_out = shared_memory.SharedMemory(name=name_in), shared_memory.SharedMemory(name=name_out)
    try:
        src = np.ndarray(shape, dtype=np.float32, buffer=seg_in.buf)
        dst = np.ndarray(shape, dtype=np.float32, buffer=seg_out.buf)
        dst[a:b] = _denoise_batched(device, src[a:b], float(sigma), BM4DProfile(**prof).native(), 2,
                                    (0.0, float(max_count)))
        del src, dst
    finally:
        seg_in.close()
        seg_out.close()
    return b - a


def denoise_patches(raw, sigma, max_count=65535.0, profile=None, device=None, devices=None):
    """``teacher = np.clip(bm4d(raw, sigma), 0, max_count)`` for a batch ``raw[N, Z, Y, X]`` of
    offset-subtracted float32 patches, in one device call (reference call pattern:
    data_handling.py:331-333 inside scripts/precompute.py:215-228).

    ``devices``: "all" or a list of device indices -- the batch is cut into one consecutive share per
    entry and every share runs in a FRESH child process that owns that device (started with the ``spawn``
    method: nothing is forked after HIP has been initialised, and this process never touches a GPU for the
    call); the patches travel through shared memory and come back in order.  Patches are independent units
    (no halo, no collective) and each carries its own fixed-point unit, so the result equals the
    single-device call bit for bit."""
    raw = np.asarray(raw, dtype=np.float32)
    if raw.ndim == 3:
        raw = raw[None]
    prof = profile or BM4DProfile()
    if devices is None:
        return _denoise_batched(device, raw, float(sigma), prof.native(), 2, (0.0, float(max_count)))
    if device is not None:
        raise ValueError("denoise_patches: give either device or devices")
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError("devices must be 'all' or a list of device indices")
        devices = list(range(max(1, _native.device_count_no_init())))
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("devices is empty")
    import multiprocessing
    from multiprocessing import shared_memory
    shares = split_batch(raw.shape[0], len(devices))
    seg_in = shared_memory.SharedMemory(create=True, size=max(1, raw.nbytes))
    seg_out = shared_memory.SharedMemory(create=True, size=max(1, raw.nbytes))
    try:
        np.ndarray(raw.shape, dtype=np.float32, buffer=seg_in.buf)[...] = raw
        jobs = [(seg_in.name, seg_out.name, raw.shape, a, b, float(sigma), float(max_count), asdict(prof), d)
                for (a, b), d in zip(shares, devices)]
        mp = multiprocessing.get_context("spawn")
        with mp.Pool(len(jobs)) as pool:
            done = pool.map(_device_share, jobs, chunksize=1)
        if sum(done) != raw.shape[0]:
            raise RuntimeError("denoise_patches: a device share did not complete")
        return np.ndarray(raw.shape, dtype=np.float32, buffer=seg_out.buf).copy()
    finally:
        for seg in (seg_in, seg_out):
            seg.close()
            seg.unlink()


def denoise_volume(vol_u16, sigma, offset=0.0, profile=None, stages=2, device=None):
    """uint16 volume -> uint16 volume: ``(float)v - offset`` -> BM4D -> ``+ offset`` -> clip to
    [0, 65535] -> rint -> uint16, entirely on the device (read_counts + bm4d + clip + the
    rint/uint16 cast of IntensityTransform.inverse).  The uint16 form matches its second stage on the basic
    estimate rounded to counts: both matching passes are 16-bit integer work."""
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
