"""The two volumes on which the HIP pipeline used to differ from the oracle (rounds 2 and 3) -- now
regression vectors for "no difference at all".

Rounds 1-3 aggregated in floating point (fp64 LDS ring, fp32 global atomics, fp32 corner weights), so
the last bits of the basic estimate depended on the order in which waves and workgroups added; stage
2's match tables are discontinuous in those bits, and a changed group moved uint16 results by a
count on a few voxels per thousand -- by 2-4 counts next to strong edges
(tests/golden/fuzz_tie_volume.npz, round 2, structure + isolated 0 / 65535 voxels, sigma 24, offset
100.5: 3.6 % of the stage-2 groups changed; tests/golden/fuzz_sparse_volume.npz, round 3,
tools/fuzz_parity.py seed 11 iteration 220: noise with one 9059-count box, sigma 16: one changed group,
0.09 % of the voxels 2-4 counts off).  Round 3 pinned an ACCOUNT of those differences; round 4 removed
their cause (integer sums): basic estimate, stage-2 tables, fp32 result and uint16 volume
equal the oracle's bit for bit, on every launch."""
import os

import numpy as np
import pytest

from aind_exaspim_image_compression import _native

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _keys(ctx, vol, sigma, c_match):
    g = [len(_native.grid_positions(n)) for n in vol.shape]
    d_vol, d_keys = ctx.to_device(vol), ctx.alloc(g[0] * g[1] * g[2] * 64)
    try:
        ctx.blockmatch(d_vol, vol.shape, sigma, c_match, d_keys)
        ctx.sync()
        return d_keys.download((g[0], g[1], g[2], 16), np.uint32)
    finally:
        d_vol.free()
        d_keys.free()


@pytest.mark.parametrize("name,sigma,offset", [("fuzz_tie_volume.npz", 24.0, 100.5),
                                               ("fuzz_sparse_volume.npz", 16.0, 0.0)])
def test_the_volumes_that_used_to_differ_are_bit_exact_now(ctx, oracle, name, sigma, offset):
    vol = np.load(os.path.join(HERE, "golden", name))["vol"]
    shape, n = vol.shape, vol.size
    f = vol.astype(np.float32) - np.float32(offset)
    E = oracle.U16_DATA_EXP
    basic_o = oracle.bm4d(f, sigma, stages=1, data_exp=E)
    want = oracle.bm4d_u16(vol, sigma, offset)
    # the GPU's own basic estimate (staged entry points with the uint16 pipelines' unit) ...
    keys1 = _keys(ctx, f, sigma, 3.0)
    np.testing.assert_array_equal(keys1, oracle.blockmatch(f, sigma, 3.0))
    d_f, d_k = ctx.to_device(f), ctx.to_device(keys1)
    d_num, d_den, d_est = ctx.alloc(4 * n), ctx.alloc(4 * n), ctx.alloc(4 * n)
    try:
        ctx.stage(d_f, None, d_k, shape, sigma, d_num, d_den, data_exp=E)
        ctx.normalize(d_num, d_den, d_est, n)
        ctx.sync()
        basic_g = d_est.download(shape, np.float32)
    finally:
        for b in (d_f, d_k, d_num, d_den, d_est):
            b.free()
    np.testing.assert_array_equal(basic_g, basic_o)
    # ... hence the same stage-2 tables (the step that used to diverge), on the estimate itself (fp32 form) and
    # on its counts (uint16 form) ...
    np.testing.assert_array_equal(_keys(ctx, basic_g, sigma, 0.6), oracle.blockmatch(basic_o, sigma, 0.6))
    counts_o = oracle.round_counts(basic_o, offset)
    d_b = ctx.to_device(basic_g)
    try:
        ctx.round_counts(d_b, d_b, n, float(offset))
        ctx.sync()
        counts_g = d_b.download(shape, np.float32)
    finally:
        d_b.free()
    np.testing.assert_array_equal(counts_g, counts_o)
    np.testing.assert_array_equal(_keys(ctx, counts_g, sigma, 0.6), oracle.blockmatch(counts_o, sigma, 0.6))
    # ... and the same uint16 volume, launch after launch
    d_u, d_o = ctx.to_device(vol), ctx.alloc(vol.nbytes)
    try:
        for _ in range(3):
            ctx.denoise_u16(d_u, d_o, shape, sigma, offset)
            ctx.sync()
            np.testing.assert_array_equal(d_o.download(shape, np.uint16), want)
    finally:
        d_u.free()
        d_o.free()
