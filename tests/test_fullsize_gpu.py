"""BASELINE.json full size (1024^3) through size-independent properties, plus the bm4d() shim and
the single-process slab driver.  GPU."""
import numpy as np
import pytest
import torch

from util import psnr, synth_volume

pytestmark = pytest.mark.gpu
SIGMA = 24.0


def test_bm4d_shim_call_signature(oracle):
    """`from bm4d import bm4d; bm4d(raw, sigma)` as at reference data_handling.py:332, and the
    evaluator form `np.maximum(bm4d(noise, 10), 0).astype(int)` on a 54^3 crop (evaluate.py:202)."""
    from bm4d import bm4d
    raw, _ = synth_volume((64, 64, 64), seed=31)
    teacher = np.clip(bm4d(raw, SIGMA), 0, 65535.0)
    want = np.clip(oracle.bm4d(raw, SIGMA), 0, 65535.0)
    assert teacher.dtype == np.float32 and psnr(teacher, want, 1000.0) > 80.0
    crop = raw[5:-5, 5:-5, 5:-5]
    gt = np.maximum(bm4d(crop, 10), 0).astype(int)
    want = np.maximum(oracle.bm4d(np.ascontiguousarray(crop), 10.0), 0).astype(int)
    assert np.mean(gt != want) < 1e-3 and np.abs(gt - want).max() <= 1
    from aind_exaspim_image_compression.bm4d import denoise_patches
    batch = np.stack([raw, raw[::-1].copy()])
    t2 = denoise_patches(batch, SIGMA)
    assert t2.shape == batch.shape and t2.min() >= 0.0
    assert psnr(t2[0], teacher, 1000.0) > 80.0
    # sub-batching (one device call per patch here) gives the same teachers
    import aind_exaspim_image_compression.bm4d as B
    old = B._MAX_VOXELS_PER_CALL
    B._MAX_VOXELS_PER_CALL = 64 ** 3
    try:
        t3 = denoise_patches(batch, SIGMA)
    finally:
        B._MAX_VOXELS_PER_CALL = old
    assert psnr(t3, t2, 1000.0) > 80.0


def test_slab_driver_single_gpu(oracle):
    """distributed.SlabDenoiser (staged C-ABI calls on torch tensors) == whole pipeline."""
    from aind_exaspim_image_compression.distributed import SlabDenoiser, denoise_slab, plan_slabs
    vol, _ = synth_volume((48, 40, 44), seed=33)
    plan = plan_slabs(48, 1, 0)
    den = SlabDenoiser(vol.shape, SIGMA, "cuda:0")
    out = denoise_slab(torch.from_numpy(vol).cuda(), plan, SIGMA, den.stage1, den.stage2)
    want = oracle.bm4d(vol, SIGMA)
    assert psnr(out.cpu().numpy(), want, 1000.0) > 80.0


def test_two_stage_parity_on_a_multi_tile_volume(ctx, oracle):
    """112 x 100 x 108 uint16, both stages: dozens of tiles, several z-layers per workgroup chunk,
    ragged extents on two axes (clamped last grid points) -- the whole device pipeline (block
    matching, half-group stage kernels, denominator convolution, uint16 rounding) against the
    oracle: at most one count off on a handful of voxels, same PSNR to 1e-3 dB."""
    shape = (112, 100, 108)
    vol, clean = synth_volume(shape, seed=77, as_u16=True)
    want = oracle.bm4d_u16(vol, 24.0, 37.0, stages=2)
    d_in = ctx.to_device(vol)
    d_out = ctx.alloc(vol.nbytes)
    try:
        ctx.denoise_u16(d_in, d_out, shape, 24.0, 37.0, stages=2)
        ctx.sync()
        got = d_out.download(shape, np.uint16)
    finally:
        d_in.free()
        d_out.free()
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 2 and np.mean(d > 0) < 1e-3 and np.mean(d > 1) < 1e-5
    peak = float(clean.max() - clean.min())
    assert abs(psnr(got, clean + 37, peak) - psnr(want, clean + 37, peak)) < 1e-3


def test_full_size_1024_properties(ctx):
    """1024^3 uint16 (BASELINE.json configs[2]): (a) locality / crop invariance -- the interior of
    a separately denoised 256^3 crop whose origin is a multiple of 4 equals the same voxels of
    the full result (dependency radius 48); (b) the removed residual has the noise's variance."""
    import bench
    n = 1024
    vol = bench.synth_u16((n, n, n), seed=5)
    d_in = ctx.to_device(vol)
    d_out = ctx.alloc(vol.nbytes)
    ctx.denoise_u16(d_in, d_out, (n, n, n), SIGMA, bench.OFFSET)
    ctx.sync()
    full = d_out.download((n, n, n), np.uint16)
    d_in.free()
    d_out.free()
    o = (384, 512, 300)
    crop = np.ascontiguousarray(vol[o[0]:o[0] + 256, o[1]:o[1] + 256, o[2]:o[2] + 256])
    from aind_exaspim_image_compression.bm4d import denoise_volume
    sub = denoise_volume(crop, SIGMA, offset=bench.OFFSET)
    a = sub[48:-48, 48:-48, 48:-48].astype(np.int32)
    b = full[o[0] + 48:o[0] + 208, o[1] + 48:o[1] + 208, o[2] + 48:o[2] + 208].astype(np.int32)
    assert np.abs(a - b).max() <= 1 and np.mean(a != b) < 1e-3
    resid = full[::8, ::8, ::8].astype(np.float32) - vol[::8, ::8, ::8].astype(np.float32)
    assert 0.8 * SIGMA < resid.std() < 1.05 * SIGMA
    assert abs(resid.mean()) < 0.5
