"""Why does the HIP two-stage pipeline differ from the oracle by a count on a few voxels per
thousand, and up to 2 % on pathological volumes?  Pinned here on the volume that tripped
tools/fuzz_parity.py in round 2 (structure + isolated 0 / 65535 voxels;
tests/golden/fuzz_tie_volume.npz), with sigma 24 and offset 100.5:

(a) Given the SAME basic estimate, stage 2 on the GPU equals the oracle: match tables bit for bit,
    estimates within the fp32 summation tolerance, uint16 equal except on rounding near-ties.
(b) In the pipeline the GPU's basic estimate differs from the oracle's in its last bits
    (aggregation order), so a few stage-2 block distances cross the admission bound or swap
    places in the top 16: a few match tables differ, and with them the groups.  Every voxel where
    the two uint16 results differ is either on a rounding near-tie of the oracle's fp32 estimate
    or inside the aggregation footprint of a reference block whose stage-2 table differs.
The round-2 note that blamed offsets with fraction .5 was wrong (tools/dbg/tie_probe.py: offsets 0,
37, 100.5 and 36.73 differ alike); tools/fuzz_parity.py's bound is the one asserted in (b)."""
import os

import numpy as np
import pytest

from aind_exaspim_image_compression import _native

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SIGMA, OFFSET = 24.0, 100.5


def _grid(shape):
    return [len(_native.grid_positions(n)) for n in shape]


def _keys(ctx, vol, c_match):
    g = _grid(vol.shape)
    d_vol, d_keys = ctx.to_device(vol), ctx.alloc(g[0] * g[1] * g[2] * 64)
    try:
        ctx.blockmatch(d_vol, vol.shape, SIGMA, c_match, d_keys)
        ctx.sync()
        return d_keys.download((g[0], g[1], g[2], 16), np.uint32)
    finally:
        d_vol.free()
        d_keys.free()


def _to_u16(pre):
    return np.rint(np.clip(pre, 0, 65535)).astype(np.int64)


def _tie_distance(pre):
    return np.abs(pre - np.floor(pre) - np.float32(0.5))


def test_differences_come_from_stage2_tables_not_from_rounding_of_half_offsets(ctx, oracle):
    vol = np.load(os.path.join(HERE, "golden", "fuzz_tie_volume.npz"))["vol"]
    shape = vol.shape
    n = vol.size
    f = vol.astype(np.float32) - np.float32(OFFSET)
    basic_o = oracle.bm4d(f, SIGMA, stages=1).astype(np.float32)
    pre_o = oracle.bm4d(f, SIGMA, stages=2).astype(np.float32) + np.float32(OFFSET)
    want = _to_u16(pre_o)
    assert np.array_equal(want, oracle.bm4d_u16(vol, SIGMA, OFFSET).astype(np.int64))

    # (a) stage 2 from the oracle's basic estimate
    keys_o = oracle.blockmatch(basic_o, SIGMA, 0.6)
    np.testing.assert_array_equal(_keys(ctx, basic_o, 0.6), keys_o)
    d_f, d_b, d_k = ctx.to_device(f), ctx.to_device(basic_o), ctx.to_device(keys_o)
    d_num, d_den, d_est = ctx.alloc(4 * n).zero(), ctx.alloc(4 * n).zero(), ctx.alloc(4 * n)
    ctx.stage(d_f, d_b, d_k, shape, SIGMA, d_num, d_den)
    ctx.normalize(d_num, d_den, d_est, n)
    ctx.sync()
    pre_a = d_est.download(shape, np.float32) + np.float32(OFFSET)
    for b in (d_f, d_b, d_k, d_num, d_den, d_est):
        b.free()
    # fp32 sums in another order: relative to the largest magnitude a voxel's footprint can hold
    assert np.max(np.abs(pre_a - pre_o)) <= 2e-5 * 65535.0
    diff_a = _to_u16(pre_a) != want
    assert np.mean(diff_a) < 1e-3
    assert np.all(_tie_distance(pre_o)[diff_a] <= 2e-5 * 65535.0)

    # (b) the pipeline: the GPU's own basic estimate
    d_in, d_out = ctx.to_device(f), ctx.alloc(4 * n)
    ctx.denoise_f32(d_in, d_out, shape, SIGMA, stages=1)
    ctx.sync()
    basic_g = d_out.download(shape, np.float32)
    d_in.free()
    d_out.free()
    assert np.max(np.abs(basic_g - basic_o)) <= 2e-5 * 65535.0          # last bits only
    keys_g = _keys(ctx, basic_g, 0.6)
    # a key is (quantised distance | displacement code): the last bits of the basic estimate move many
    # quantised distances by one unit without touching the group; what changes a group is its list of
    # displacement codes (members and their order -- the Haar transform runs along it)
    moved = np.any(keys_g != keys_o, axis=-1)
    changed = np.any((keys_g & 0x7FF) != (keys_o & 0x7FF), axis=-1)     # per reference block
    print(f"stage-2 keys that moved: {moved.mean():.1%} of the tables; groups that changed: {changed.mean():.2%}")
    assert 0 < changed.mean() < 0.25, changed.mean()
    d_u, d_o = ctx.to_device(vol), ctx.alloc(vol.nbytes)
    ctx.denoise_u16(d_u, d_o, shape, SIGMA, OFFSET)
    ctx.sync()
    got = d_o.download(shape, np.uint16).astype(np.int64)
    d_u.free()
    d_o.free()
    diff = got != want
    assert np.abs(got - want).max() <= 1 and diff.mean() < 2e-2, (np.abs(got - want).max(), diff.mean())
    # aggregation footprint of the changed references: blocks at displacements -5..5 of an 8^3 block
    foot = np.zeros(shape, bool)
    pz, py, px = (_native.grid_positions(m) for m in shape)
    for iz, iy, ix in zip(*np.nonzero(changed)):
        z, y, x = int(pz[iz]), int(py[iy]), int(px[ix])
        foot[max(0, z - 5):z + 13, max(0, y - 5):y + 13, max(0, x - 5):x + 13] = True
    unexplained = diff & ~foot & (_tie_distance(pre_o) > 2e-5 * 65535.0)
    print(f"stage-2 tables that differ: {changed.mean():.3%}; differing voxels {diff.mean():.3%}, of which "
          f"{np.mean(foot[diff]):.1%} inside the footprint of a changed table; unexplained {int(unexplained.sum())}")
    assert not unexplained.any()
    assert np.mean(foot[diff]) > 0.5 and foot.mean() < 0.9            # the footprint is not "everywhere"


def test_one_changed_group_next_to_a_bright_box_moves_voxels_by_several_counts(ctx, oracle):
    """The volume that tripped tools/fuzz_parity.py (seed 11, iteration 220) in round 3: N(40, 24) noise with
    one 9 x 11 x 7 box of 9059 counts, sigma 16, offset 0.  ONE stage-2 group of ~1000 differs between GPU and
    oracle (the last bits of the basic estimate moved a candidate across the admission bound); its blocks
    straddle the box's edge, so 0.09 % of the voxels move by 2 .. 4 counts.  The account of the test above
    holds: same basic estimate -> identical tables; every differing voxel lies in the footprint of the changed
    group or on a rounding near-tie.  (tools/fuzz_parity.py applies this account whenever its plain bound fails.)"""
    vol = np.load(os.path.join(HERE, "golden", "fuzz_sparse_volume.npz"))["vol"]
    sigma, shape, n = 16.0, vol.shape, vol.size
    f = vol.astype(np.float32)
    basic_o = oracle.bm4d(f, sigma, stages=1).astype(np.float32)
    pre_o = oracle.bm4d(f, sigma, stages=2).astype(np.float32)
    want = _to_u16(pre_o)
    g = _grid(shape)

    def keys_of(basic):
        d_vol, d_keys = ctx.to_device(basic), ctx.alloc(g[0] * g[1] * g[2] * 64)
        try:
            ctx.blockmatch(d_vol, shape, sigma, 0.6, d_keys)
            ctx.sync()
            return d_keys.download((g[0], g[1], g[2], 16), np.uint32)
        finally:
            d_vol.free()
            d_keys.free()

    keys_o = oracle.blockmatch(basic_o, sigma, 0.6)
    np.testing.assert_array_equal(keys_of(basic_o), keys_o)
    d_in, d_out = ctx.to_device(f), ctx.alloc(4 * n)
    ctx.denoise_f32(d_in, d_out, shape, sigma, stages=1)
    ctx.sync()
    basic_g = d_out.download(shape, np.float32)
    d_in.free()
    d_out.free()
    changed = np.any((keys_of(basic_g) & 0x7FF) != (keys_o & 0x7FF), axis=-1)
    d_u, d_o = ctx.to_device(vol), ctx.alloc(vol.nbytes)
    ctx.denoise_u16(d_u, d_o, shape, sigma, 0.0)
    ctx.sync()
    got = d_o.download(shape, np.uint16).astype(np.int64)
    d_u.free()
    d_o.free()
    diff = got != want
    foot = np.zeros(shape, bool)
    pz, py, px = (_native.grid_positions(m) for m in shape)
    for iz, iy, ix in zip(*np.nonzero(changed)):
        z, y, x = int(pz[iz]), int(py[iy]), int(px[ix])
        foot[max(0, z - 5):z + 13, max(0, y - 5):y + 13, max(0, x - 5):x + 13] = True
    unexplained = diff & ~foot & (_tie_distance(pre_o) > 2e-5 * 65535.0)
    print(f"groups changed {int(changed.sum())} of {changed.size}; max|d| {int(np.abs(got - want).max())}; "
          f"differing {diff.mean():.2e}; unexplained {int(unexplained.sum())}")
    assert not unexplained.any()
    assert changed.sum() <= 0.01 * changed.size and diff.mean() < 2e-2
    assert np.abs(got - want).max() <= 8          # bounded by the contrast the changed group's blocks straddle
