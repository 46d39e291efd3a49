"""Block matching's march (bm_kernels.hip, round 3): a workgroup walks up a segment of blocks of eight
cell layers and carries the top layer's cell sums into the next block, instead of recomputing every
eighth layer.  The same sums enter the same adds: tables must be bit-identical to the one-tile-per-
workgroup launch and to the oracle, for every segment length, with ragged last blocks, idle waves,
batches, both tile shapes, the float and the integer kernel."""
import numpy as np
import pytest

from util import synth_volume

from aind_exaspim_image_compression import _native

pytestmark = pytest.mark.gpu
SIGMA = 24.0


def _keys(ctx, vol, c_match, integer, batch=1):
    shape = vol.shape[-3:]
    g = [len(_native.grid_positions(n)) for n in shape]
    src = vol if integer else (vol.astype(np.float32) - np.float32(37.0))
    d_vol = ctx.to_device(np.ascontiguousarray(src))
    d_keys = ctx.alloc(batch * g[0] * g[1] * g[2] * 16 * 4)
    try:
        if integer:
            ctx.blockmatch_u16(d_vol, shape, SIGMA, c_match, d_keys, batch=batch)
        else:
            ctx.blockmatch(d_vol, shape, SIGMA, c_match, d_keys, batch=batch)
        ctx.sync()
        return d_keys.download((batch, *g, 16) if batch > 1 else (*g, 16), np.uint32)
    finally:
        d_vol.free()
        d_keys.free()


@pytest.fixture
def march(ctx):
    def set_march(n):
        ctx.set_option("bm_march", n)
    yield set_march
    ctx.set_option("bm_march", 1)


@pytest.mark.parametrize("integer", [False, True])
@pytest.mark.parametrize("shape", [(100, 40, 44), (72, 36, 68), (134, 24, 32)])
def test_marched_tables_equal_tiled_tables_and_the_oracle(ctx, oracle, march, shape, integer):
    """24 / 17 / 32 reference layers: segments of 2 blocks (15 layers) leave a second segment with a
    ragged last block; 3 and 8 blocks cover the volume in one segment with idle top waves; an odd
    plane count adds the clamped last position (generic kernel)."""
    vol = synth_volume(shape, seed=sum(shape), as_u16=True)[0]
    march(0)
    tiled = _keys(ctx, vol, 3.0, integer)
    if integer:
        f = vol.astype(np.float32)           # blockmatch_u16 matches on the counts themselves (offset 0)
    else:
        f = vol.astype(np.float32) - np.float32(37.0)
    np.testing.assert_array_equal(tiled, oracle.blockmatch(f, SIGMA, 3.0))
    for n in (2, 3, 8):
        march(n)
        np.testing.assert_array_equal(_keys(ctx, vol, 3.0, integer), tiled, err_msg=f"bm_march={n}")
    march(2)
    np.testing.assert_array_equal(_keys(ctx, vol, 0.6, integer), oracle.blockmatch(f, SIGMA, 0.6))


@pytest.mark.parametrize("integer", [False, True])
def test_marched_batch_of_patches_and_unaligned_planes(ctx, oracle, march, integer):
    """64^3 patches take the 4 x 16 tile shape: 15 reference layers = one segment of two blocks; a
    102-plane volume has a clamped last grid position next to its marched layers."""
    vols = np.stack([synth_volume((64, 64, 64), seed=70 + i, as_u16=True)[0] for i in range(3)])
    march(2)
    got = _keys(ctx, vols, 3.0, integer, batch=3)
    for i in range(3):
        f = vols[i].astype(np.float32) - (np.float32(0.0) if integer else np.float32(37.0))
        np.testing.assert_array_equal(got[i], oracle.blockmatch(f, SIGMA, 3.0))
    vol = synth_volume((102, 32, 40), seed=9, as_u16=True)[0]
    f = vol.astype(np.float32) - (np.float32(0.0) if integer else np.float32(37.0))
    for n in (2, 4):
        march(n)
        np.testing.assert_array_equal(_keys(ctx, vol, 3.0, integer), oracle.blockmatch(f, SIGMA, 3.0))


def test_pipeline_with_forced_march_equals_the_default(ctx, march):
    """Whole uint16 pipeline with the march forced on a small volume: the result of the default launch
    (same tables; the stage kernels' global fp32 adds are not ordered, so near-ties may round apart)."""
    from aind_exaspim_image_compression.bm4d import denoise_volume
    vol = synth_volume((96, 48, 56), seed=33, as_u16=True)[0]
    march(0)
    want = denoise_volume(vol, SIGMA, 37.0)
    march(3)
    d = np.abs(denoise_volume(vol, SIGMA, 37.0).astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and np.mean(d > 0) < 2e-3, (int(d.max()), float(np.mean(d > 0)))
