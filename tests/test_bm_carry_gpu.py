"""Block matching's carry between the tiles of a column (bm_kernels.hip, round 3): a tile advances by eight
cell layers instead of seven and takes the lower cells of its first reference layer from the tile below,
through global memory, in slab order behind a per-column counter.  The same sums enter the same adds:
tables must be bit-identical to the launch without the carry and to the oracle -- ragged last tiles, idle
waves, both tile shapes, the float and the integer kernel, batches."""
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
def carry(ctx):
    def set_carry(n):
        ctx.set_option("bm_carry", n)      # 0 off, 1 automatic (large launches), 2 forced
    yield set_carry
    ctx.set_option("bm_carry", 1)


@pytest.mark.parametrize("integer", [False, True])
@pytest.mark.parametrize("shape", [(100, 40, 44), (72, 36, 68), (134, 24, 32)])
def test_carried_tables_equal_tiled_tables_and_the_oracle(ctx, oracle, carry, shape, integer):
    """24 / 17 / 32 reference layers = 4 / 3 / 5 tiles of eight cell layers in a column, the last one with
    idle waves (1 / 2 / 1 cell layers in use); few tiles per slab, so most slab positions of an XCD are padding."""
    vol = synth_volume(shape, seed=sum(shape), as_u16=True)[0]
    carry(0)
    tiled = _keys(ctx, vol, 3.0, integer)
    if integer:
        f = vol.astype(np.float32)           # blockmatch_u16 matches on the counts themselves (offset 0)
    else:
        f = vol.astype(np.float32) - np.float32(37.0)
    np.testing.assert_array_equal(tiled, oracle.blockmatch(f, SIGMA, 3.0))
    carry(2)
    for rep in range(3):          # the slots of a column are reused by every second tile and by every launch
        np.testing.assert_array_equal(_keys(ctx, vol, 3.0, integer), tiled, err_msg=f"launch {rep}")
    np.testing.assert_array_equal(_keys(ctx, vol, 0.6, integer), oracle.blockmatch(f, SIGMA, 0.6))


@pytest.mark.parametrize("integer", [False, True])
def test_carry_with_patches_batches_and_unaligned_planes(ctx, oracle, carry, integer):
    """A 64^3 patch takes the 4 x 16 tile shape (15 reference layers = two tiles with the carry); in a batch
    the columns of all elements form one slab (3 x 5 columns here); a 102-plane volume has a clamped last
    grid position (generic kernel) next to its carried layers."""
    vols = np.stack([synth_volume((64, 64, 64), seed=70 + i, as_u16=True)[0] for i in range(3)])
    carry(2)
    f0 = vols[0].astype(np.float32) - (np.float32(0.0) if integer else np.float32(37.0))
    np.testing.assert_array_equal(_keys(ctx, vols[0], 3.0, integer), oracle.blockmatch(f0, SIGMA, 3.0))
    got = _keys(ctx, vols, 3.0, integer, batch=3)
    for i in range(3):
        f = vols[i].astype(np.float32) - (np.float32(0.0) if integer else np.float32(37.0))
        np.testing.assert_array_equal(got[i], oracle.blockmatch(f, SIGMA, 3.0))
    vol = synth_volume((102, 32, 40), seed=9, as_u16=True)[0]
    f = vol.astype(np.float32) - (np.float32(0.0) if integer else np.float32(37.0))
    np.testing.assert_array_equal(_keys(ctx, vol, 3.0, integer), oracle.blockmatch(f, SIGMA, 3.0))


def test_pipeline_with_forced_carry_equals_the_default(ctx, carry):
    """Whole uint16 pipeline with the carry forced on a small volume: the result of the launch without it
    (same tables, integer aggregation sums: the same uint16 volume)."""
    from aind_exaspim_image_compression.bm4d import denoise_volume
    vol = synth_volume((96, 48, 56), seed=33, as_u16=True)[0]
    carry(0)
    want = denoise_volume(vol, SIGMA, 37.0)
    carry(2)
    np.testing.assert_array_equal(denoise_volume(vol, SIGMA, 37.0), want)


def test_large_launch_takes_the_carry_by_default_and_agrees(ctx, carry):
    """128 x 640 x 640: 23 x 23 tiles per slab, 4 slabs with the carry (31 reference layers) against 5
    without; the automatic setting (1) must give the tables of the launch without it, from both kernels."""
    rng = np.random.default_rng(12)
    vol = np.clip(rng.normal(37.0, SIGMA, (128, 640, 640)), 0, 65535).round().astype(np.uint16)
    vol[30:90, 100:300, 200:420] += 400
    got = {}
    for mode in (0, 1):
        carry(mode)
        got[mode] = (_keys(ctx, vol, 3.0, True), _keys(ctx, vol, 0.6, False))
    np.testing.assert_array_equal(got[0][0], got[1][0])
    np.testing.assert_array_equal(got[0][1], got[1][1])


@pytest.mark.parametrize("integer", [False, True])
def test_large_batch_of_patches_takes_the_carry_by_default(ctx, carry, integer):
    """120 patches of 64^3 (the shape of scripts/precompute.py's work): 600 columns in one slab, two tiles per
    column with the carry instead of three -- the automatic setting against the launch without it, every
    element; and the chunk-local mode's batches of padded chunks through the whole pipeline."""
    rng = np.random.default_rng(5)
    base = np.stack([synth_volume((64, 64, 64), seed=200 + i, as_u16=True)[0] for i in range(6)])
    vols = np.ascontiguousarray(base[rng.integers(0, 6, 120)])
    vols += rng.integers(0, 3, vols.shape, dtype=np.uint16)            # 120 different patches
    carry(0)
    want = _keys(ctx, vols, 3.0, integer, batch=120)
    carry(1)
    np.testing.assert_array_equal(_keys(ctx, vols, 3.0, integer, batch=120), want)


def test_chunk_local_mode_with_and_without_the_carry(ctx, carry):
    from aind_exaspim_image_compression.bm4d import denoise_chunked
    vol = synth_volume((72, 200, 200), seed=41, as_u16=True)[0]        # 2 x 5 x 5 cores of 40 + 8: 14 reference layers
    carry(0)
    want = denoise_chunked(vol, SIGMA, 37.0, chunk=40, halo=8)
    carry(2)
    np.testing.assert_array_equal(denoise_chunked(vol, SIGMA, 37.0, chunk=40, halo=8), want)


def test_a_carry_wait_that_runs_out_is_an_error_not_a_hang(ctx, oracle, carry):
    """Round 4: tiles take their place in the launch order by ticket, so a tile's producer has always started
    and the wait ends; it is bounded all the same.  The debug option "bm_carry_fault" makes every wait count
    as run out: the launch finishes (void tables), the next synchronising call reports EXABM4D_ERR_HIP and
    names the carry, the context switches the carry off and works -- and the bm4d() path
    (exabm4d_denoise_f32_host) repeats its run without the carry by itself."""
    vol = synth_volume((100, 40, 44), seed=5, as_u16=True)[0]
    f = vol.astype(np.float32) - np.float32(37.0)
    want = oracle.blockmatch(f, SIGMA, 3.0)
    carry(2)
    assert _native.blockmatch_plan(vol.shape, ctx=ctx)["carry"]
    ctx.set_option("bm_carry_fault", 1)
    try:
        with pytest.raises(_native.NativeError, match="carry"):
            _keys(ctx, vol, 3.0, False)                       # ctx.sync() inside sees the status word
        assert not _native.blockmatch_plan(vol.shape, ctx=ctx)["carry"]          # off for this context now
        np.testing.assert_array_equal(_keys(ctx, vol, 3.0, False), want)         # and the context works
        carry(2)                                              # forced again, fault still armed: the host entry recovers
        got = ctx.denoise_f32_host(f, SIGMA, stages=1)
        np.testing.assert_array_equal(got, oracle.bm4d(f, SIGMA, stages=1))
        assert not _native.blockmatch_plan(vol.shape, ctx=ctx)["carry"]
    finally:
        ctx.set_option("bm_carry_fault", 0)
    carry(2)
    np.testing.assert_array_equal(_keys(ctx, vol, 3.0, False), want)             # the carry itself is intact


def test_options_belong_to_their_context(ctx):
    """Round 4 (ADVICE): bm_carry / bm_xcd_mode / stage_* used to be process globals behind a per-context
    setter.  A second context on the same device keeps the defaults whatever the first one sets."""
    other = _native.Context(0)
    try:
        ctx.set_option("bm_carry", 0)
        assert not _native.blockmatch_plan((1024, 1024, 1024), ctx=ctx)["carry"]
        assert _native.blockmatch_plan((1024, 1024, 1024), ctx=other)["carry"]
        assert _native.blockmatch_plan((1024, 1024, 1024))["carry"]
        ctx.set_option("bm_xcd_mode", 0)
        assert _native.blockmatch_plan((64, 640, 640), ctx=ctx)["slab_order_q"] == 0
        assert _native.blockmatch_plan((64, 640, 640), ctx=other)["slab_order_q"] > 0
    finally:
        ctx.set_option("bm_carry", 1)
        ctx.set_option("bm_xcd_mode", 2)
        other.close()
