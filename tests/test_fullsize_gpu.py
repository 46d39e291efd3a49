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
    assert teacher.dtype == np.float32
    np.testing.assert_array_equal(teacher, want)
    crop = raw[5:-5, 5:-5, 5:-5]
    gt = np.maximum(bm4d(crop, 10), 0).astype(int)
    want = np.maximum(oracle.bm4d(np.ascontiguousarray(crop), 10.0), 0).astype(int)
    np.testing.assert_array_equal(gt, want)
    from aind_exaspim_image_compression.bm4d import denoise_patches
    batch = np.stack([raw, raw[::-1].copy()])
    t2 = denoise_patches(batch, SIGMA)
    assert t2.shape == batch.shape and t2.min() >= 0.0
    np.testing.assert_array_equal(t2[0], teacher)          # a patch of a batch == the patch alone
    # sub-batching (one device call per patch here) gives the same teachers
    import aind_exaspim_image_compression.bm4d as B
    old = B._MAX_VOXELS_PER_CALL
    B._MAX_VOXELS_PER_CALL = 64 ** 3
    try:
        t3 = denoise_patches(batch, SIGMA)
    finally:
        B._MAX_VOXELS_PER_CALL = old
    np.testing.assert_array_equal(t3, t2)


def test_slab_driver_single_gpu(oracle):
    """distributed.SlabDenoiser (staged C-ABI calls on torch tensors) == whole pipeline."""
    from aind_exaspim_image_compression.distributed import SlabDenoiser, denoise_slab, plan_slabs
    vol, _ = synth_volume((48, 40, 44), seed=33)
    plan = plan_slabs(48, 1, 0)
    den = SlabDenoiser(vol.shape, SIGMA, "cuda:0")
    out = denoise_slab(torch.from_numpy(vol).cuda(), plan, SIGMA, den.stage1, den.stage2)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.bm4d(vol, SIGMA))


def test_two_stage_parity_on_a_multi_tile_volume(ctx, oracle):
    """112 x 100 x 108 uint16, both stages: dozens of tiles, several z-layers per workgroup chunk,
    ragged extents on two axes (clamped last grid points) -- the whole device pipeline (block
    matching, half-group stage kernels, denominator convolution, uint16 rounding) against the
    oracle: the same uint16 volume."""
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
    np.testing.assert_array_equal(got, want)


@pytest.fixture(scope="module")
def full1024(ctx):
    """(noisy, denoised) 1024^3 uint16 volumes of BASELINE.json configs[2] / [4], host arrays."""
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
    return vol, full


def test_full_size_1024_properties(full1024):
    """1024^3 uint16 (BASELINE.json configs[2]): (a) locality / crop invariance -- the interior of
    a separately denoised 256^3 crop whose origin is a multiple of 4 equals the same voxels of
    the full result (dependency radius 48); (b) the removed residual has the noise's variance."""
    import bench
    vol, full = full1024
    o = (384, 512, 300)
    crop = np.ascontiguousarray(vol[o[0]:o[0] + 256, o[1]:o[1] + 256, o[2]:o[2] + 256])
    from aind_exaspim_image_compression.bm4d import denoise_volume
    sub = denoise_volume(crop, SIGMA, offset=bench.OFFSET)
    # integer aggregation sums and a fixed unit: the same blocks add the same integers in crop and volume
    np.testing.assert_array_equal(sub[48:-48, 48:-48, 48:-48],
                                  full[o[0] + 48:o[0] + 208, o[1] + 48:o[1] + 208, o[2] + 48:o[2] + 208])
    resid = full[::8, ::8, ::8].astype(np.float32) - vol[::8, ::8, ::8].astype(np.float32)
    assert 0.8 * SIGMA < resid.std() < 1.05 * SIGMA
    assert abs(resid.mean()) < 0.5


def _oracle_crop_check(oracle, vol, got, origin, edge, radius, stages):
    """The oracle on a crop whose origin is a multiple of 4 reproduces the full-size result on the
    crop's interior (everything further than the dependency radius from the crop's faces)."""
    o, e, r = origin, edge, radius
    assert all(v % 4 == 0 for v in o)
    crop = np.ascontiguousarray(vol[o[0]:o[0] + e, o[1]:o[1] + e, o[2]:o[2] + e])
    want = oracle.bm4d_u16(crop, SIGMA, 37.0, stages=stages)[r:e - r, r:e - r, r:e - r]
    have = got[o[0] + r:o[0] + e - r, o[1] + r:o[1] + e - r, o[2] + r:o[2] + e - r]
    np.testing.assert_array_equal(have, want, err_msg=f"crop at {origin}")


def test_config3_1024_against_the_oracle_on_interior_crops(full1024, oracle):
    """BASELINE.json configs[2] (1024^3, two stages) anchored on the ORACLE: two 144^3 crops at
    different depths, interiors of 48^3 voxels (dependency radius 48 for two stages): the oracle's
    uint16 values, voxel for voxel."""
    vol, full = full1024
    for origin in ((400, 516, 128), (40, 860, 700)):
        _oracle_crop_check(oracle, vol, full, origin, 144, 48, 2)


def test_config2_256_against_the_oracle_on_interior_crops(ctx, oracle):
    """BASELINE.json configs[1] (256^3, hard-threshold stage + aggregation) against the oracle on
    two interior crops (radius 24 for one stage), and the two-stage result of the same volume on
    one crop."""
    import bench
    vol = bench.synth_u16((256, 256, 256), seed=9)
    d_in = ctx.to_device(vol)
    d_out = ctx.alloc(vol.nbytes)
    try:
        ctx.denoise_u16(d_in, d_out, vol.shape, SIGMA, 37.0, stages=1)
        ctx.sync()
        ht = d_out.download(vol.shape, np.uint16)
        ctx.denoise_u16(d_in, d_out, vol.shape, SIGMA, 37.0, stages=2)
        ctx.sync()
        two = d_out.download(vol.shape, np.uint16)
    finally:
        d_in.free()
        d_out.free()
    for origin in ((0, 0, 0), (112, 60, 144)):              # one crop shares three volume faces
        _oracle_crop_check(oracle, vol, ht, origin, 112, 24, 1)
    _oracle_crop_check(oracle, vol, two, (64, 128, 20), 128, 48, 2)


def test_config5_chain_at_1024(full1024, ctx):
    """BASELINE.json configs[4] at full size: denoised volume -> (a) lossless chunk coder, decode
    == input; (b) 8^3 block DCT quantiser -> chunk coder on the indices -> decode == indices ->
    dequantise: error bounded by the step.  Three chunk streams are compared with the oracle's
    bytes (EXAC v2, the default); the coded sizes lie below the order-0 entropy of the byte planes
    (v1's floor), i.e. the context model pays."""
    from aind_exaspim_image_compression import _native
    from oracle import codec_oracle as co
    vol, full = full1024
    n = 1024
    shape = (n, n, n)
    dev = torch.device("cuda", 0)
    t_full = torch.from_numpy(full.view(np.int16)).to(dev)
    ctx.sync()

    def encode(t_src, ts, vshape, chunk):
        nchunks = int(np.prod([-(-a // c) for a, c in zip(vshape, chunk)]))
        cap = _native.codec_volume_bound(ts, vshape, chunk)
        t_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        t_off = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
        t_sz = torch.empty(nchunks, dtype=torch.int32, device=dev)
        exact, container = ctx.codec_encode(t_src, ts, vshape, chunk, out=t_out, out_capacity=cap,
                                            offsets=t_off, sizes=t_sz)
        assert container <= cap and int(t_sz.sum()) == exact
        return t_out, t_off, t_sz, exact

    # (a) lossless leg on the denoised uint16 volume, 64^3 chunks
    t_out, t_off, t_sz, exact = encode(t_full, 2, shape, (64, 64, 64))
    t_back = torch.empty_like(t_full)
    ctx.codec_decode(t_out, t_out.numel(), t_off, 2, shape, (64, 64, 64), t_back)
    assert torch.equal(t_back, t_full)
    off, sz = t_off.cpu().numpy(), t_sz.cpu().numpy()
    for c in (0, 1234, 4095):
        cz, cy, cx = c // 256, (c // 16) % 16, c % 16
        chunk = full[64 * cz:64 * cz + 64, 64 * cy:64 * cy + 64, 64 * cx:64 * cx + 64]
        want = co.encode(chunk)
        got = t_out[int(off[c]):int(off[c]) + int(sz[c])].cpu().numpy().tobytes()
        assert got == want, f"chunk {c}"
        assert len(want) < co.plane_entropy_bytes(chunk)     # below what any order-0 byte-plane coder can reach
    assert 4.0 < 2.0 * n ** 3 / exact < 20.0                 # denoised data compress; raw ~2
    del t_out, t_back
    # (b) config 5's lossy leg
    q = 8.0
    nblk = (n // 8) ** 3
    t_idx = torch.empty(nblk * 512, dtype=torch.int32, device=dev)
    ctx.dctq_forward(t_full, shape, q, t_idx)
    ishape, ichunk = (nblk, 8, 64), (512, 8, 64)
    t_out, t_off, t_sz, exact_i = encode(t_idx, 4, ishape, ichunk)
    t_iback = torch.empty_like(t_idx)
    ctx.codec_decode(t_out, t_out.numel(), t_off, 4, ishape, ichunk, t_iback)
    assert torch.equal(t_iback, t_idx)
    first = t_idx[:1 << 18].cpu().numpy().reshape(ichunk)
    o0, s0 = int(t_off[0]), int(t_sz[0])
    assert t_out[o0:o0 + s0].cpu().numpy().tobytes() == co.encode(first)
    t_rec = torch.empty_like(t_full)
    ctx.dctq_inverse(t_iback, shape, q, t_rec)
    ctx.sync()
    err = (t_rec.to(torch.int32) & 0xFFFF) - (t_full.to(torch.int32) & 0xFFFF)
    assert int(err.abs().max()) <= 0.5 * q * np.sqrt(512.0) + 1.0
    assert float(err.float().abs().mean()) < 0.5 * q
    assert exact_i < exact                                    # the lossy leg is the smaller one


def test_config3_bm4dnet_stage_at_1024(full1024):
    """BASELINE.json configs[2] is "two-stage BM4D + bm4dnet learned shrinkage" on a 1024^3 volume: the
    learned stage -- inference.predict (reference inference.py:28-116; production call
    scripts/evaluate_bm4dnet.py:136, :201) -- at the FULL size on the BM4D result: 8000 patches of
    64^3, batch 32.  No oracle exists at this size, so the checks are properties: uint16 volume of
    the input's shape; the reference's low-edge quirk (first `trim` voxels of every axis =
    transform.inverse(0) = the offset, inference.py:91-103); and locality -- a 256^3 crop whose origin
    lies on the patch grid (multiples of 52) predicted on its own gives the same voxels wherever
    the same patches with the same content cover them (8 <= p < 212 per axis).  Run for the seeded
    U-Net (fp32, random init: throughput and plumbing, not quality) and for an elementwise model,
    whose result is also known in closed form."""
    from aind_exaspim_image_compression import inference
    from aind_exaspim_image_compression.machine_learning import transforms as T
    from aind_exaspim_image_compression.machine_learning import unet3d
    _, den = full1024
    n = 1024
    cfg = {"kind": "offset", "base": {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}},
           "params": {"offset": 37.0}}
    tf = T.build_transform(cfg)
    o = (520, 312, 208)
    assert all(v % 52 == 0 for v in o)
    crop = np.ascontiguousarray(den[o[0]:o[0] + 256, o[1]:o[1] + 256, o[2]:o[2] + 256])
    inner = (slice(8, 212),) * 3
    full_inner = tuple(slice(a + 8, a + 212) for a in o)

    class Affine(torch.nn.Module):
        def forward(self, x):
            return x * 0.5 + 0.125

    # (a) elementwise model: closed form + locality
    out = inference.predict(den, Affine().cuda().eval(), tf, batch_size=32, verbose=False)
    assert out.dtype == np.uint16 and out.shape == (n, n, n)
    assert np.all(out[:5] == 37) and np.all(out[:, :5] == 37) and np.all(out[:, :, :5] == 37)
    sub = inference.predict(crop, Affine().cuda().eval(), tf, batch_size=32, verbose=False)
    np.testing.assert_array_equal(sub[inner], out[full_inner])
    want = tf.inverse(tf.forward(crop[inner]) * np.float32(0.5) + np.float32(0.125))
    d = np.abs(out[full_inner].astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and np.mean(d > 0) < 1e-3          # mean of k equal fp32 values, k = 1 .. 8 patches
    del out
    # (b) the U-Net
    torch.manual_seed(0)
    model = unet3d.UNet().cuda().eval()
    inference.predict(den[:64, :220, :428], model, tf, batch_size=32, verbose=False)    # one full batch: MIOpen warm-up
    out = inference.predict(den, model, tf, batch_size=32, verbose=False)
    assert out.dtype == np.uint16 and out.shape == (n, n, n)
    assert np.all(out[:5] == 37) and np.all(out[:, :5] == 37) and np.all(out[:, :, :5] == 37)
    assert inference.count_patches(inference._ShapeOnly((1, 1, n, n, n)), 64, 12) == 8000
    sub = inference.predict(crop, model, tf, batch_size=32, verbose=False)
    d = np.abs(sub[inner].astype(np.int32) - out[full_inner].astype(np.int32))
    # same patches, same fp32 network; a patch may sit at another position of its batch
    assert d.max() <= 1 and np.mean(d > 0) < 1e-3, (int(d.max()), float(np.mean(d > 0)))
    assert out[full_inner].std() > 0                        # a real volume came back, not a constant
