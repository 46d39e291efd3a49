"""GPU parity: BM4D kernels (through the C-ABI) vs the CPU oracle on identical inputs.

Bars: since round 4 EVERYTHING is bit-exact.  Match tables always were; the aggregation's sums are
64-bit integers now and the Wiener weights use a bit-defined reciprocal (3.7), so the
stage outputs, the fp32 estimates and the uint16 volumes equal the oracle's bit for bit, whatever the
order in which waves, workgroups and launches add (rounds 1-3: fp32 tolerance, uint16 within a count,
2-4 counts on rare voxels after a changed stage-2 group).
"""
import numpy as np
import pytest

from util import psnr, synth_volume

from aind_exaspim_image_compression import _native

pytestmark = pytest.mark.gpu

SIGMA = 24.0


def _keys_gpu(ctx, vol, sigma, c_match):
    from aind_exaspim_image_compression import _native as nat
    g = [len(nat.grid_positions(n)) for n in vol.shape]
    d_vol = ctx.to_device(vol)
    d_keys = ctx.alloc(g[0] * g[1] * g[2] * 16 * 4)
    ctx.blockmatch(d_vol, vol.shape, sigma, c_match, d_keys)
    ctx.sync()
    return d_keys.download((g[0], g[1], g[2], 16), np.uint32)


@pytest.mark.parametrize("shape", [(64, 64, 64), (40, 44, 48), (8, 8, 8), (16, 12, 36),
                                   (54, 54, 54), (30, 37, 41), (9, 10, 11)])
def test_blockmatch_bit_exact(ctx, oracle, shape):
    vol, _ = synth_volume(shape, seed=3)
    want = oracle.blockmatch(vol, SIGMA, 3.0)
    got = _keys_gpu(ctx, vol, SIGMA, 3.0)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("shape", [(64, 64, 64), (54, 54, 54), (8, 8, 8), (16, 12, 36),
                                   (30, 37, 41), (9, 10, 100)])
def test_blockmatch_on_guarded_copy_bit_exact(ctx, oracle, shape):
    """The pipelines match on the library's own buffers, where x-edge tiles skip the per-element
    clamping and read past the row ends (poisoned with NaN patterns here): same match tables."""
    vol, _ = synth_volume(shape, seed=sum(shape) + 1)
    want = oracle.blockmatch(vol, SIGMA, 3.0)
    ctx.set_option("bm_guarded_copy", 1)
    try:
        got = _keys_gpu(ctx, vol, SIGMA, 3.0)
        batch = np.stack([vol, vol[::-1].copy()])
        g = got.shape[:3]
        d_vol = ctx.to_device(batch)
        d_keys = ctx.alloc(2 * g[0] * g[1] * g[2] * 16 * 4)
        ctx.blockmatch(d_vol, vol.shape, SIGMA, 3.0, d_keys, batch=2)
        ctx.sync()
        both = d_keys.download((2,) + got.shape, np.uint32)
    finally:
        ctx.set_option("bm_guarded_copy", 0)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(both[0], want)
    np.testing.assert_array_equal(both[1], oracle.blockmatch(batch[1], SIGMA, 3.0))


def test_blockmatch_generic_kernel_matches(ctx, oracle):
    """The one-wave-per-block kernel alone must also reproduce the oracle bit for bit."""
    vol, _ = synth_volume((36, 40, 44), seed=5)
    want = oracle.blockmatch(vol, SIGMA, 3.0)
    ctx.set_option("force_generic_bm", 1)
    try:
        got = _keys_gpu(ctx, vol, SIGMA, 3.0)
    finally:
        ctx.set_option("force_generic_bm", 0)
    np.testing.assert_array_equal(got, want)


def test_blockmatch_tables_are_never_empty(ctx, oracle):
    """blocks holding an infinity or a NaN get the one-entry table [0] (their distance to
    themselves is not a number), all other tables are unchanged -- in both float kernels, as in the oracle."""
    vol, _ = synth_volume((40, 44, 48), seed=6)
    vol[5, 6, 7] = np.nan
    vol[30, 33, 41] = -np.inf
    vol[39, 43, 47] = np.inf
    want = oracle.blockmatch(vol, SIGMA, 3.0)
    assert (want[..., 0] == 0).all() and (want[1, 1, 1, 1:] == 0xFFFFFFFF).all()
    np.testing.assert_array_equal(_keys_gpu(ctx, vol, SIGMA, 3.0), want)
    ctx.set_option("force_generic_bm", 1)
    try:
        np.testing.assert_array_equal(_keys_gpu(ctx, vol, SIGMA, 3.0), want)
    finally:
        ctx.set_option("force_generic_bm", 0)


def test_blockmatch_constant_volume_self_first(ctx, oracle):
    """All candidates tie at distance 0: the reference block itself must still be entry 0 and the
    rest ordered by displacement code."""
    vol = np.full((24, 24, 24), 5.0, dtype=np.float32)
    got = _keys_gpu(ctx, vol, SIGMA, 3.0)
    want = oracle.blockmatch(vol, SIGMA, 3.0)
    np.testing.assert_array_equal(got, want)
    assert np.all(got[..., 0] == 0)


def test_blockmatch_wiener_threshold(ctx, oracle):
    vol, _ = synth_volume((32, 32, 32), seed=7, sigma=4.0)
    want = oracle.blockmatch(vol, SIGMA, 0.6)
    got = _keys_gpu(ctx, vol, SIGMA, 0.6)
    np.testing.assert_array_equal(got, want)


def _stage_gpu(ctx, noisy, keys, sigma, basic=None, data_exp=None):
    d_noisy = ctx.to_device(noisy)
    d_basic = ctx.to_device(basic) if basic is not None else None
    d_keys = ctx.to_device(keys)
    d_num = ctx.alloc(noisy.nbytes).fill(0xFF)        # the call WRITES both (NaN patterns must not survive)
    d_den = ctx.alloc(noisy.nbytes).fill(0xFF)
    try:
        ctx.stage(d_noisy, d_basic, d_keys, noisy.shape, sigma, d_num, d_den, data_exp=data_exp)
        ctx.sync()
        return d_num.download(noisy.shape, np.float32), d_den.download(noisy.shape, np.float32)
    finally:
        for b in (d_noisy, d_basic, d_keys, d_num, d_den):
            if b is not None:
                b.free()


def _assert_stage_is_the_oracles(ctx, oracle, noisy, keys, basic=None, port=False, data_exp=None):
    num_w, den_w = oracle.stage(noisy, keys, SIGMA, basic=basic, port=port, data_exp=data_exp)
    num_g, den_g = _stage_gpu(ctx, noisy, keys, SIGMA, basic=basic, data_exp=data_exp)
    assert np.all(den_g > 0)
    np.testing.assert_array_equal(den_g, den_w)
    np.testing.assert_array_equal(num_g, num_w)
    return num_g, den_g


@pytest.mark.parametrize("shape", [(40, 44, 48), (30, 37, 41)])
def test_hard_threshold_stage(ctx, oracle, shape):
    noisy, _ = synth_volume(shape, seed=11)
    keys = oracle.blockmatch(noisy, SIGMA, 3.0)
    _assert_stage_is_the_oracles(ctx, oracle, noisy, keys)
    _assert_stage_is_the_oracles(ctx, oracle, noisy, keys, data_exp=17)       # the uint16 pipelines' unit


def _mixed_volume(shape, seed):
    """Smooth background (groups of 16), a band of high-contrast white noise (groups of one or two
    blocks: nothing matches), and a gradient in between (intermediate group sizes)."""
    rng = np.random.default_rng(seed)
    vol = rng.normal(100.0, SIGMA, shape).astype(np.float32)
    y0, y1 = shape[1] // 3, 2 * shape[1] // 3
    vol[:, y0:y1, :] += rng.uniform(0, 6000, (shape[0], y1 - y0, shape[2])).astype(np.float32)
    vol[:, :, : shape[2] // 4] += np.linspace(0, 900, shape[2] // 4, dtype=np.float32)
    return vol


def test_marching_tiles_with_every_group_size_equal_the_cpu_port(ctx, oracle):
    """A volume large enough that tiles march over several z-layers and z chunks, with every group size
    present (one-block groups leave the second wave of a pair idle): both stage kernels against the CPU
    port (bit-identical to the oracle, tests/test_oracle_bm4d.py) on the WHOLE volume, bit for bit, and
    twice in a row -- the second launch adds in another order."""
    shape = (48, 192, 200)
    noisy = _mixed_volume(shape, 23)
    keys = _keys_gpu(ctx, noisy, SIGMA, 3.0)
    sizes = np.unique((keys != 0xFFFFFFFF).sum(axis=-1))
    assert sizes.min() <= 1 and sizes.max() == 16
    a = _assert_stage_is_the_oracles(ctx, oracle, noisy, keys, port=True)
    b = _stage_gpu(ctx, noisy, keys, SIGMA)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    basic = (noisy + np.random.default_rng(5).normal(0, 2.0, shape)).astype(np.float32)   # any second volume will do
    _assert_stage_is_the_oracles(ctx, oracle, noisy, keys, basic=basic, port=True)


def test_tall_ragged_volume_every_launch_shape_gives_the_oracles_result(ctx, oracle):
    """Regression (round 2): tile columns that march over MANY layers in one z chunk, with edge
    tiles that hold fewer groups than the workgroup has wave pairs -- a pair without a group in a
    layer must not let a layer complete (and its ring planes be flushed) before the layer below.
    One chunk of 63 layers, the automatic chunking and 9 chunks: the CPU port's uint16 volume each time."""
    import bench
    shape = (253, 61, 57)
    vol = bench.synth_u16(shape, 7)
    vol[:40] = 0                                  # zero padding: every candidate matches
    vol[:, :, -9:] = 0
    want = oracle.bm4d_u16(vol, SIGMA, 37.0, port=True)
    d_in = ctx.to_device(vol)
    d_out = ctx.alloc(vol.nbytes)
    try:
        for chunks in (1, 0, 9):
            ctx.set_option("stage_chunks", chunks)
            ctx.denoise_u16(d_in, d_out, shape, SIGMA, 37.0)
            ctx.sync()
            np.testing.assert_array_equal(d_out.download(shape, np.uint16), want, err_msg=f"stage_chunks {chunks}")
    finally:
        ctx.set_option("stage_chunks", 0)
        d_in.free()
        d_out.free()


def test_wiener_stage(ctx, oracle):
    shape = (40, 44, 48)
    noisy, _ = synth_volume(shape, seed=13)
    basic = oracle.bm4d(noisy, SIGMA, stages=1)
    keys = oracle.blockmatch(basic, SIGMA, 0.6)
    _assert_stage_is_the_oracles(ctx, oracle, noisy, keys, basic=basic)


@pytest.mark.parametrize("stages", [1, 2])
def test_pipeline_f32_psnr(ctx, oracle, stages):
    shape = (64, 64, 64)
    noisy, clean = synth_volume(shape, seed=17)
    want = oracle.bm4d(noisy, SIGMA, stages=stages)
    got = ctx.denoise_f32_host(noisy, SIGMA, stages=stages)
    np.testing.assert_array_equal(got, want)
    peak = float(clean.max() - clean.min())
    assert psnr(got, clean, peak) > psnr(noisy, clean, peak) + 8.0        # it actually denoises


def test_pipeline_batch_of_patches(ctx, oracle):
    """N independent 48^3 patches in one call == N single calls (precompute.py call pattern)."""
    vols = np.stack([synth_volume((48, 48, 48), seed=s)[0] for s in (1, 2, 3)])
    got = ctx.denoise_f32_host(vols, SIGMA, clip=(0.0, 65535.0))
    for i in range(3):       # (every volume of a batch gets its own numerator unit, like a single call)
        np.testing.assert_array_equal(got[i], np.clip(oracle.bm4d(vols[i], SIGMA), 0, 65535))


def test_pipeline_u16(ctx, oracle):
    shape = (48, 52, 56)
    vol, _ = synth_volume(shape, seed=19, as_u16=True)
    want = oracle.bm4d_u16(vol, SIGMA, 37.0)
    d_in = ctx.to_device(vol)
    d_out = ctx.alloc(vol.nbytes)
    ctx.denoise_u16(d_in, d_out, shape, SIGMA, 37.0)
    ctx.sync()
    np.testing.assert_array_equal(d_out.download(shape, np.uint16), want)
    ctx.denoise_u16(d_in, d_out, shape, SIGMA, 37.0)           # and again: a deterministic function of its input
    ctx.sync()
    np.testing.assert_array_equal(d_out.download(shape, np.uint16), want)


def test_host_batch_of_scattered_volumes(ctx, oracle):
    """exabm4d_denoise_f32_host_v: one pointer per volume (the broker's call, every worker's patch in its own
    segment): per volume the result of the contiguous host call, in place and out of place."""
    shape = (24, 28, 32)
    vols = [synth_volume(shape, seed=70 + i)[0] * np.float32(1 + i) for i in range(5)]
    want = ctx.denoise_f32_host(np.stack(vols), SIGMA, clip=(0.0, 65535.0))
    np.testing.assert_array_equal(want[3], np.clip(oracle.bm4d(vols[3], SIGMA), 0, 65535))
    outs = [np.full(shape, -1, dtype=np.float32) for _ in vols]
    ctx.denoise_f32_host_v([v.ctypes.data for v in vols], [o.ctypes.data for o in outs], shape, SIGMA,
                           clip=(0.0, 65535.0))
    for i in range(5):
        np.testing.assert_array_equal(outs[i], want[i])
    work = [v.copy() for v in vols]
    addrs = [w.ctypes.data for w in work]
    ctx.denoise_f32_host_v(addrs, addrs, shape, SIGMA, clip=(0.0, 65535.0))
    for i in range(5):
        np.testing.assert_array_equal(work[i], want[i])
    with pytest.raises(ValueError):
        ctx.denoise_f32_host_v(addrs[:2], addrs[:1], shape, SIGMA)


def test_large_host_batch_in_overlapped_sub_batches(ctx, oracle):
    """exabm4d_denoise_f32_host cuts a batch of >= 2^27 voxels into sub-batches of 2^26 and copies under the
    kernels (option ``host_pipeline``): 520 patches of 64^3 = 256 + 256 + 8.  Same bits as the call in one
    piece and as the oracle (spot checks: first, a middle one, the ragged tail)."""
    base = np.stack([synth_volume((64, 64, 64), seed=80 + i)[0] for i in range(8)])
    raw = np.concatenate([base + np.float32(3 * k) for k in range(65)])
    assert raw.shape[0] == 520
    got = ctx.denoise_f32_host(raw, SIGMA, clip=(0.0, 65535.0))
    ctx.set_option("host_pipeline", 0)
    try:
        one = ctx.denoise_f32_host(raw, SIGMA, clip=(0.0, 65535.0))
    finally:
        ctx.set_option("host_pipeline", 1)
    np.testing.assert_array_equal(got, one)
    for i in (0, 300, 519):
        np.testing.assert_array_equal(got[i], np.clip(oracle.bm4d(raw[i], SIGMA), 0, 65535))


def test_calls_queued_back_to_back_and_the_zeroing_stream(ctx, oracle):
    """The 8-byte sums of a large call (>= 2^25 voxels) are zeroed on a second stream under block matching
    (option ``zero_overlap``).  Calls queued without a synchronisation between them share the
    sums' memory: the zeroing of call k + 1 must wait for the last reader of call k.  Same bits in line and
    overlapped, large and small calls mixed, and the oracle's (its CPU port: 34 M voxels)."""
    big, small = (132, 504, 512), (40, 44, 48)
    assert np.prod(big) >= 1 << 25 > np.prod(small)
    vols = [np.tile(synth_volume((132, 126, 128), seed=61, as_u16=True)[0], (1, 4, 4)),
            synth_volume(small, seed=62, as_u16=True)[0]]
    vols.append(vols[0][::-1].copy())
    want_small = oracle.bm4d_u16(vols[1], SIGMA, 37.0)
    want_big = oracle.bm4d_u16(vols[0], SIGMA, 37.0, port=True)
    d_in = [ctx.to_device(v) for v in vols]
    d_out = [ctx.alloc(v.nbytes) for v in vols]
    got = {}
    for overlap in (1, 0):
        ctx.set_option("zero_overlap", overlap)
        try:
            for d in d_out:
                d.fill(0xEE)
            for i in range(3):
                ctx.denoise_u16(d_in[i], d_out[i], vols[i].shape, SIGMA, 37.0)
            ctx.sync()
        finally:
            ctx.set_option("zero_overlap", 1)
        got[overlap] = [d_out[i].download(vols[i].shape, np.uint16) for i in range(3)]
    for i in range(3):
        np.testing.assert_array_equal(got[1][i], got[0][i])
    np.testing.assert_array_equal(got[1][0], want_big)
    np.testing.assert_array_equal(got[1][1], want_small)
    for d in d_in + d_out:
        d.free()


@pytest.mark.parametrize("shape,offset", [((29, 37, 42), 37.0), ((45, 9, 68), 37.0), ((33, 30, 41), 37.0),
                                          ((29, 37, 42), 36.73), ((40, 44, 48), 100.5)])
def test_pipeline_u16_stage2_matches_on_counts_in_every_kernel(ctx, oracle, shape, offset):
    """stage 2 of the uint16 pipeline matches on the basic estimate rounded to counts -- in the
    integer kernel (even rows, an offset that is exact in fp32), in the one-wave kernel for reference blocks at
    clamped grid positions (extents - 8 that are no multiple of 4: it reads the same counts as fp32), in the
    float kernel otherwise (odd rows, offset 36.73), and with every kernel forced off in turn: one result."""
    vol, _ = synth_volume(shape, seed=sum(shape), as_u16=True)
    vol[::7, ::5, ::3] = 65535                       # clamped ends of the count range
    vol[3::7, 2::5, 1::3] = 0
    want = oracle.bm4d_u16(vol, SIGMA, offset)
    d_in, d_out = ctx.to_device(vol), ctx.alloc(vol.nbytes)
    try:
        for option, value in ((None, 0), ("bm_int", 0), ("force_generic_bm", 1)):
            if option:
                ctx.set_option(option, value)
            try:
                d_out.fill(0xEE)
                ctx.denoise_u16(d_in, d_out, shape, SIGMA, offset)
                ctx.sync()
            finally:
                if option:
                    ctx.set_option(option, 1 - value)
            np.testing.assert_array_equal(d_out.download(shape, np.uint16), want)
    finally:
        d_in.free()
        d_out.free()


@pytest.mark.parametrize("shape", [(8, 8, 8), (8, 9, 12), (12, 8, 8), (9, 9, 9), (16, 8, 20),
                                   (8, 64, 8), (13, 11, 10)])
def test_tiny_and_thin_volumes_full_pipeline(ctx, oracle, shape):
    """One reference block per axis, one-tile volumes, clamped grid points on every axis: most
    wave pairs of a workgroup have nothing to do, the rest must still synchronise correctly."""
    vol = synth_volume(shape, seed=sum(shape))[0]
    for stages in (1, 2):
        want = oracle.bm4d(vol, SIGMA, stages=stages)
        got = ctx.denoise_f32_host(vol, SIGMA, stages=stages)
        np.testing.assert_array_equal(got, want)


def test_bad_arguments_raise(ctx):
    from aind_exaspim_image_compression import _native as nat
    with pytest.raises(ValueError):
        ctx.denoise_f32_host(np.zeros((4, 8, 8), np.float32), SIGMA)       # axis < 8
    with pytest.raises(ValueError):
        ctx.denoise_f32_host(np.zeros((8, 8, 8), np.float32), -1.0)        # sigma <= 0
    with pytest.raises(ValueError):
        ctx.denoise_f32_host(np.zeros((8, 8, 8), np.float32), SIGMA,
                             params=nat.default_params(block=4))           # unsupported profile


def test_zero_padded_volume_and_small_groups(ctx, oracle):
    """Edge cases of the domain: a volume that is mostly zero padding (every candidate of a block
    ties at distance 0, aggregation sees all-zero groups), a bright block with few matches
    (group sizes 1, 2, 4, 8 appear) -- match tables and both stages bit-exact."""
    rng = np.random.default_rng(23)
    vol = np.zeros((40, 36, 44), dtype=np.float32)
    vol[12:28, 10:26, 14:34] = rng.normal(300.0, SIGMA, (16, 16, 20)).astype(np.float32)
    vol[20:23, 15:18, 20:23] += 20000.0
    keys = _keys_gpu(ctx, vol, SIGMA, 3.0)
    want = oracle.blockmatch(vol, SIGMA, 3.0)
    np.testing.assert_array_equal(keys, want)
    counts = (want != 0xFFFFFFFF).sum(-1)
    assert counts.min() < 16 and counts.max() == 16
    num_g, den_g = _assert_stage_is_the_oracles(ctx, oracle, vol, want)
    basic = oracle.normalize(num_g, den_g)
    keys2 = oracle.blockmatch(basic, SIGMA, 0.6)
    np.testing.assert_array_equal(_keys_gpu(ctx, basic, SIGMA, 0.6), keys2)
    _assert_stage_is_the_oracles(ctx, oracle, vol, keys2, basic=basic)


def test_batch_of_unaligned_volumes_stage1_u16(ctx, oracle):
    """batch > 1 with extents that are not 8 (mod 4) (clamped last grid point on every axis) and
    the hard-threshold-only uint16 pipeline."""
    shape = (22, 27, 33)
    vols = np.stack([synth_volume(shape, seed=s, as_u16=True)[0] for s in (41, 42)])
    d_in = ctx.to_device(vols)
    d_out = ctx.alloc(vols.nbytes)
    ctx.denoise_u16(d_in, d_out, shape, SIGMA, 37.0, stages=1, batch=2)
    ctx.sync()
    got = d_out.download(vols.shape, np.uint16)
    for i in range(2):
        np.testing.assert_array_equal(got[i], oracle.bm4d_u16(vols[i], SIGMA, 37.0, stages=1))


def test_other_sigma_and_profile(ctx, oracle):
    """sigma in other units (normalised data) and a non-default profile (beta = 0, lambda, c)."""
    from aind_exaspim_image_compression import _native as nat
    vol, clean = synth_volume((32, 32, 40), seed=29)
    small = (vol / np.float32(4000.0)).astype(np.float32)
    got = ctx.denoise_f32_host(small, SIGMA / 4000.0)
    want = oracle.bm4d(small, SIGMA / 4000.0)
    np.testing.assert_array_equal(got, want)
    big = (vol * np.float32(3.0e7)).astype(np.float32)         # ... and data far above the uint16 range (E = 38)
    np.testing.assert_array_equal(ctx.denoise_f32_host(big, SIGMA * 3.0e7), oracle.bm4d(big, SIGMA * 3.0e7))
    for k in (-60, 40):        # the unit follows the data over 100 binades (E = -46 ... 54)
        v = (vol * np.float32(2.0 ** k)).astype(np.float32)
        assert oracle.data_exp(v) == oracle.data_exp(vol) + k
        np.testing.assert_array_equal(ctx.denoise_f32_host(v, SIGMA * 2.0 ** k), oracle.bm4d(v, SIGMA * 2.0 ** k))
    p = nat.default_params(kaiser_beta=0.0, lambda_ht=3.0, c_match_ht=2.5, c_match_wie=0.4)
    got = ctx.denoise_f32_host(vol, SIGMA, params=p)
    want = oracle.bm4d(vol, SIGMA, kaiser_beta=0.0, lambda_ht=3.0, c_match_ht=2.5, c_match_wie=0.4)
    np.testing.assert_array_equal(got, want)


def test_fp32_input_outside_the_working_range_is_refused(ctx, oracle):
    """DESIGN.md 3.8, domain: |v| >= 2^56 (squares of transform coefficients leave fp32), infinities and NaNs
    are reported as EXABM4D_ERR_INVALID at the call's synchronisation -- by the device, which finds the
    largest magnitude anyway; the oracle's wrapper refuses the same volumes.  The context stays usable."""
    vol, _ = synth_volume((16, 24, 24), seed=21)
    big = (vol * np.float32(2.0 ** (57 - oracle.data_exp(vol)))).astype(np.float32)       # E = 57
    assert oracle.data_exp(big) > oracle.MAX_DATA_EXP
    nan = vol.copy()
    nan[3, 4, 5] = np.nan
    inf = vol.copy()
    inf[0, 0, 0] = -np.inf
    for bad in (big, nan, inf):
        with pytest.raises(ValueError, match="working range"):
            ctx.denoise_f32_host(bad, SIGMA)
        with pytest.raises(ValueError, match="working range"):
            oracle.bm4d(bad, SIGMA)
    edge = (vol * np.float32(2.0 ** (56 - oracle.data_exp(vol)))).astype(np.float32)      # E = 56: the last one in
    assert oracle.data_exp(edge) == 56
    s = SIGMA * 2.0 ** (56 - oracle.data_exp(vol))
    np.testing.assert_array_equal(ctx.denoise_f32_host(edge, s), oracle.bm4d(edge, s))
    np.testing.assert_array_equal(ctx.denoise_f32_host(vol, SIGMA), oracle.bm4d(vol, SIGMA))


def _keys_u16(ctx, vol, sigma, c_match, batch=1):
    shape = vol.shape[-3:]
    g = [len(_native.grid_positions(n)) for n in shape]
    d_vol = ctx.to_device(vol)
    d_keys = ctx.alloc(batch * g[0] * g[1] * g[2] * 16 * 4)
    try:
        ctx.blockmatch_u16(d_vol, shape, sigma, c_match, d_keys, batch=batch)
        ctx.sync()
        return d_keys.download((batch, g[0], g[1], g[2], 16) if batch > 1 else (g[0], g[1], g[2], 16),
                               np.uint32)
    finally:
        d_vol.free()
        d_keys.free()


@pytest.mark.parametrize("shape", [(24, 28, 32), (40, 44, 64), (26, 31, 22), (64, 64, 64), (33, 72, 130),
                                   (24, 24, 25)])
def test_integer_block_matching_equals_the_oracle(ctx, oracle, shape):
    """Stage-1 matching of the uint16 pipelines (bm_tile16_kernel: saturating int16 differences,
    v_dot2 accumulation) against the oracle on (float)v - offset: bit-exact tables, on aligned and
    ragged extents (odd nx falls back to the float kernel), with saturating voxels (0 next to
    65535: differences beyond int16), for both admission bounds, and with the integer path
    switched off."""
    vol = synth_volume(shape, seed=sum(shape), as_u16=True)[0]
    vol.reshape(-1)[:: max(1, vol.size // 23)] = 65535
    vol.reshape(-1)[5:: max(1, vol.size // 19)] = 0
    f = vol.astype(np.float32) - np.float32(37.0)
    for sigma, c_match in ((SIGMA, 3.0), (SIGMA, 0.6), (90.0, 3.0), (120.0, 3.0)):   # 120: bound > 2^24 -> float kernel
        want = oracle.blockmatch(f, sigma, c_match)
        np.testing.assert_array_equal(_keys_u16(ctx, vol, sigma, c_match), want)
    ctx.set_option("bm_int", 0)
    try:
        np.testing.assert_array_equal(_keys_u16(ctx, vol, SIGMA, 3.0), oracle.blockmatch(f, SIGMA, 3.0))
    finally:
        ctx.set_option("bm_int", 1)


def test_integer_block_matching_batch_of_patches(ctx, oracle):
    """A batch of 64^3 patches (the 4 x 16 tile shape) through the integer kernel."""
    vols = np.stack([synth_volume((64, 64, 64), seed=50 + i, as_u16=True)[0] for i in range(3)])
    got = _keys_u16(ctx, vols, SIGMA, 3.0, batch=3)
    for i in range(3):
        np.testing.assert_array_equal(got[i], oracle.blockmatch(vols[i].astype(np.float32) - 37.0, SIGMA, 3.0))


def test_non_dyadic_offset_takes_the_float_kernel(ctx, oracle):
    """An offset such as 36.73 (a percentile from estimate_offset) makes (float)v - offset inexact in
    fp32: voxel differences are no longer integers, so the integer matching kernel would not give
    the float kernel's tables.  The launcher must fall back to the float kernel for such offsets
    (offset * 128 not an integer): switching the integer path off changes nothing, and the result
    is the oracle's (which always matches on the fp32 counts).  With a dyadic offset the integer
    kernel does run, and its result is the float kernel's too.  All of it bit for bit."""
    vol = synth_volume((40, 48, 64), seed=77, as_u16=True)[0]

    def run(offset, bm_int):
        ctx.set_option("bm_int", bm_int)
        d_in, d_out = ctx.to_device(vol), ctx.alloc(vol.nbytes)
        try:
            ctx.denoise_u16(d_in, d_out, vol.shape, SIGMA, offset)
            return d_out.download(vol.shape, np.uint16)
        finally:
            ctx.set_option("bm_int", 1)
            d_in.free()
            d_out.free()

    for offset in (36.73, 0.3, 100.5, 37.0):
        a, b = run(offset, 1), run(offset, 0)
        np.testing.assert_array_equal(a, b, err_msg=f"offset {offset}: integer / float matching")
        np.testing.assert_array_equal(a, oracle.bm4d_u16(vol, SIGMA, offset), err_msg=f"offset {offset}")
    # the gate itself, on the host side used by the slab driver
    from aind_exaspim_image_compression.distributed import offset_exact_in_fp32
    assert [offset_exact_in_fp32(o) for o in (0.0, 37.0, 100.5, 0.0078125, 36.73, 0.3, 70000.0)] == \
        [True, True, True, True, False, False, False]


def test_wiener_gathers_from_the_interleaved_volume(ctx, oracle):
    """Round 3: the Wiener kernel reads block k's noisy and basic values with one 8-byte load per row
    from an interleaved (noisy, basic) volume (half the cache lines of two gathers) and runs them as
    the two packed streams of one transform -- per stream the same IEEE operations as the separate
    gathers (option stage_pairvol = 0): identical outputs, on a volume with every group size, and the
    CPU port's."""
    shape = (40, 120, 128)
    noisy = _mixed_volume(shape, 41)
    basic = (noisy + np.random.default_rng(6).normal(0, 2.0, shape)).astype(np.float32)
    keys = _keys_gpu(ctx, basic, SIGMA, 3.0)
    # every group size: cut the sorted tables to random lengths (a prefix of a table is a table)
    cut = np.random.default_rng(7).choice([1, 2, 3, 5, 8, 11, 16], size=keys.shape[:3])
    keys[np.arange(16)[None, None, None, :] >= cut[..., None]] = 0xFFFFFFFF
    assert np.unique((keys != 0xFFFFFFFF).sum(axis=-1)).size > 4
    res = {}
    try:
        for pv in (1, 0):
            ctx.set_option("stage_pairvol", pv)
            res[pv] = _stage_gpu(ctx, noisy, keys, SIGMA, basic=basic)
    finally:
        ctx.set_option("stage_pairvol", 1)
    np.testing.assert_array_equal(res[1][0], res[0][0])
    np.testing.assert_array_equal(res[1][1], res[0][1])
    num_w, den_w = oracle.stage(noisy, keys, SIGMA, basic=basic, port=True)
    np.testing.assert_array_equal(res[1][0], num_w)
    np.testing.assert_array_equal(res[1][1], den_w)


def test_block_matching_workgroup_orders_give_identical_tables(ctx):
    """bm_xcd_mode: 0 = every XCD walks its own contiguous range of tiles, 1 = all XCDs inside one z slab of
    tiles at a time (raster order, padding workgroups that exit at once), 2 (default for large launches) / 3 /
    5 = the slab in strips of that many tile rows (23 rows: ragged last strips).  64 x 640 x 640: 3 slabs of
    23 x 23 tiles, 7 padding positions per slab.  Same tables from both kernels."""
    rng = np.random.default_rng(11)
    vol = np.clip(rng.normal(37.0, SIGMA, (64, 640, 640)), 0, 65535).round().astype(np.uint16)
    vol[20:40, 100:300, 200:420] += 500
    f = vol.astype(np.float32) - np.float32(37.0)
    g = [len(_native.grid_positions(n)) for n in vol.shape]
    d_u16, d_f32 = ctx.to_device(vol), ctx.to_device(f)
    d_keys = ctx.alloc(g[0] * g[1] * g[2] * 16 * 4)
    got = {}
    try:
        for mode in (0, 1, 2, 3, 5):
            ctx.set_option("bm_xcd_mode", mode)
            ctx.blockmatch_u16(d_u16, vol.shape, SIGMA, 3.0, d_keys)
            ctx.sync()
            a = d_keys.download((*g, 16), np.uint32)
            ctx.blockmatch(d_f32, vol.shape, SIGMA, 0.6, d_keys)
            ctx.sync()
            got[mode] = (a, d_keys.download((*g, 16), np.uint32))
    finally:
        ctx.set_option("bm_xcd_mode", 2)
        for b in (d_u16, d_f32, d_keys):
            b.free()
    for mode in (1, 2, 3, 5):
        np.testing.assert_array_equal(got[0][0], got[mode][0], err_msg=f"integer kernel, mode {mode}")
        np.testing.assert_array_equal(got[0][1], got[mode][1], err_msg=f"fp32 kernel, mode {mode}")
    assert (got[1][0][..., 0] & 0x7FF).max() == 0 and (got[1][0][..., 1] != 0xFFFFFFFF).mean() > 0.5   # self first, groups found


def test_stage_tile_order_option_changes_nothing_but_the_order(ctx):
    """stage_strip = n walks the stage kernels' tile columns in strips of n tile rows (default 3: -0.4 % at
    1024^3), 0 in raster order.  Same groups, same arithmetic, integer sums: the same uint16 volume."""
    from aind_exaspim_image_compression.bm4d import denoise_volume
    vol = synth_volume((40, 150, 170), seed=17, as_u16=True)[0]
    ctx.set_option("stage_strip", 0)
    want = denoise_volume(vol, SIGMA, 37.0)
    try:
        for n in (2, 3, 7):
            ctx.set_option("stage_strip", n)
            np.testing.assert_array_equal(denoise_volume(vol, SIGMA, 37.0), want, err_msg=f"stage_strip {n}")
    finally:
        ctx.set_option("stage_strip", 3)
