"""Self-consistency of the BM4D oracle (the reference holds no test for its third-party bm4d
wheel -- parity unpinned -- so these are the properties SURVEY.md section 8c asks for).  CPU."""
import os

import numpy as np
import pytest

from util import psnr, synth_volume

SIGMA = 24.0


def test_grid_positions(oracle):
    assert oracle.grid_positions(64).tolist() == list(range(0, 57, 4))
    assert oracle.grid_positions(54).tolist() == list(range(0, 45, 4)) + [46]
    assert oracle.grid_positions(8).tolist() == [0]
    assert oracle.grid_positions(9).tolist() == [0, 1]
    assert oracle.grid_positions(7).tolist() == []
    for n, c in ((64, 15), (256, 63), (1024, 255), (272, 67)):
        assert len(oracle.grid_positions(n)) == c


def test_tables(oracle):
    dct, win = oracle.tables(2.0)
    np.testing.assert_allclose(dct @ dct.T, np.eye(8), atol=1e-6)       # orthonormal DCT-II
    k = np.kaiser(8, 2.0)
    np.testing.assert_allclose(win, k[:, None, None] * k[None, :, None] * k[None, None, :],
                               rtol=1e-6)
    _, ones = oracle.tables(0.0)
    assert np.all(ones == 1.0)


@pytest.mark.parametrize("K", [1, 2, 4, 8, 16])
def test_group_transform_parseval_and_inverse(oracle, K):
    g = np.random.default_rng(K).normal(size=(K, 8, 8, 8)).astype(np.float32)
    G = oracle.group_transform(g)
    assert abs(np.sum(G.astype(np.float64) ** 2) / np.sum(g.astype(np.float64) ** 2) - 1) < 1e-5
    np.testing.assert_allclose(oracle.group_transform(G, inverse=True), g, atol=5e-6)
    if K > 1:
        const = np.ones((K, 8, 8, 8), np.float32)
        C = oracle.group_transform(const)
        assert abs(C[0, 0, 0, 0] - np.sqrt(K * 512.0)) < 1e-2       # all energy in DC
        assert np.abs(C).sum() - abs(C[0, 0, 0, 0]) < 1e-2


def _haar_matrix(K):
    """Orthonormal Haar analysis matrix of the specification: approximations
    first, recursively; float64."""
    if K == 1:
        return np.ones((1, 1))
    c = 1.0 / np.sqrt(2.0)
    a = np.zeros((K // 2, K))
    d = np.zeros((K // 2, K))
    for i in range(K // 2):
        a[i, 2 * i] = a[i, 2 * i + 1] = c
        d[i, 2 * i], d[i, 2 * i + 1] = c, -c
    return np.vstack([_haar_matrix(K // 2) @ a, d])


@pytest.mark.parametrize("K", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("port", [False, True])
def test_group_transform_is_the_dct_haar_of_the_definition(oracle, K, port):
    """The folded 36-operation butterfly of DESIGN.md 3.5 IS the transform it claims to be: the
    oracle's (and the CPU port's) group transform against the float64 definition
    Haar_K (x) DCT-II_8 (x) DCT-II_8 (x) DCT-II_8 built from the textbook formula -- not from the
    oracle's own table (test_tables pins that separately)."""
    n = np.arange(8)
    D = np.sqrt(2.0 / 8.0) * np.cos(np.pi * (2 * n[None, :] + 1) * n[:, None] / 16.0)
    D[0] /= np.sqrt(2.0)
    g = np.random.default_rng(100 + K).normal(size=(K, 8, 8, 8)).astype(np.float32)
    want = np.einsum("kj,ua,vb,wc,jabc->kuvw", _haar_matrix(K), D, D, D, g.astype(np.float64))
    got = oracle.group_transform(g, port=port)
    assert np.max(np.abs(got - want)) < 2e-6 * max(1.0, np.sqrt(K))     # measured 6e-7
    back = oracle.group_transform(want.astype(np.float32), inverse=True, port=port)
    assert np.max(np.abs(back - g)) < 4e-6


def test_match_table_invariants(oracle):
    vol, _ = synth_volume((32, 36, 40), seed=2)
    keys = oracle.blockmatch(vol, SIGMA, 3.0)
    assert np.all(keys[..., 0] == 0)                                 # the block itself is first
    valid = keys != 0xFFFFFFFF
    body = keys.astype(np.int64)
    assert np.all((np.diff(body, axis=-1) > 0) | ~valid[..., 1:])    # strictly ascending
    assert np.all(keys[valid] < oracle.keymax(SIGMA, 3.0))
    # candidates stay inside the volume
    pz, py, px = (oracle.grid_positions(n) for n in vol.shape)
    for (iz, iy, ix) in [(0, 0, 0), (len(pz) - 1, len(py) - 1, len(px) - 1), (3, 4, 5)]:
        for (d, _) in oracle.decode_keys(keys[iz, iy, ix]):
            c = np.array([pz[iz], py[iy], px[ix]]) + np.array(d)
            assert np.all(c >= 0) and np.all(c + 8 <= np.array(vol.shape))


def test_constant_volume_is_a_fixed_point(oracle):
    vol = np.full((24, 24, 24), 123.0, dtype=np.float32)
    np.testing.assert_allclose(oracle.bm4d(vol, SIGMA), vol, rtol=1e-5)


def test_lambda_zero_is_identity(oracle):
    vol, _ = synth_volume((24, 24, 28), seed=4)
    np.testing.assert_allclose(oracle.bm4d(vol, SIGMA, stages=1, lambda_ht=0.0), vol,
                               atol=2e-3)


def test_small_sigma_returns_input(oracle):
    vol, _ = synth_volume((24, 24, 24), seed=5)
    out = oracle.bm4d(vol, 1e-3)
    assert np.abs(out - vol).max() < 1e-2


def test_psnr_gain_and_stage_order(oracle):
    noisy, clean = synth_volume((48, 48, 48), seed=6)
    peak = float(clean.max() - clean.min())
    p0 = psnr(noisy, clean, peak)
    p1 = psnr(oracle.bm4d(noisy, SIGMA, stages=1), clean, peak)
    p2 = psnr(oracle.bm4d(noisy, SIGMA, stages=2), clean, peak)
    assert p1 > p0 + 8.0
    assert p2 > p1


def test_u16_pipeline_equals_float_pipeline(oracle):
    vol, _ = synth_volume((24, 28, 32), seed=8, as_u16=True)
    # the uint16 form fixes the numerator's unit (E = 17) instead of reading it off the data, and
    # matches stage 2 on the basic estimate rounded to counts
    f = oracle.bm4d(vol.astype(np.float32) - np.float32(37.0), SIGMA, data_exp=oracle.U16_DATA_EXP,
                    match_counts_offset=37.0)
    want = np.rint(np.clip(f + np.float32(37.0), 0, 65535)).astype(np.uint16)
    np.testing.assert_array_equal(oracle.bm4d_u16(vol, SIGMA, 37.0), want)
    np.testing.assert_array_equal(oracle.bm4d_u16(vol, SIGMA, 37.0, port=True), want)
    # ... which is the pipeline spelled out: stage 1, round, match on the rounded volume, filter with the unrounded one
    x = vol.astype(np.float32) - np.float32(37.0)
    E = oracle.U16_DATA_EXP
    basic = oracle.bm4d(x, SIGMA, stages=1, data_exp=E)
    m = oracle.round_counts(basic, 37.0)
    np.testing.assert_array_equal(m, np.rint(np.clip(basic + np.float32(37.0), 0, 65535)).astype(np.float32)
                                  - np.float32(37.0))
    assert np.abs(m - basic).max() <= 0.5 + 1e-3
    keys = oracle.blockmatch(m, SIGMA, oracle.DEFAULTS["c_match_wie"])
    num, den = oracle.stage(x, keys, SIGMA, basic=basic, data_exp=E)
    np.testing.assert_array_equal(oracle.normalize(num, den), f)
    # the fp32 form matches on the basic estimate itself: a different (equally good) set of groups
    g = oracle.bm4d(x, SIGMA, data_exp=E)
    assert not np.array_equal(g, f) and np.abs(g - f).max() < 0.25 * SIGMA


def test_crop_invariance(oracle):
    """A voxel's result depends on input within 48 voxels (two stages x 24), and the reference
    grid of a crop whose origin is a multiple of 4 coincides with the volume's."""
    vol, _ = synth_volume((40, 40, 120), seed=9)
    full = oracle.bm4d(vol, SIGMA)
    crop = oracle.bm4d(vol[:, :, 4:116], SIGMA)
    np.testing.assert_allclose(crop[:, :, 48:-48], full[:, :, 52:-52], rtol=1e-4, atol=1e-3)


def test_dct_quantiser_oracle_properties(oracle):
    """DESIGN.md 3.10 on the CPU: orthonormal transform (energy preserved up to quantisation),
    step 1 reproduces the volume to within one count, edge blocks replicate the last voxel."""
    rng = np.random.default_rng(3)
    vol = np.clip(rng.normal(400, 80, (20, 17, 33)), 0, 65535).astype(np.uint16)
    idx = oracle.dctq_forward(vol, 1.0)
    assert idx.shape == (3, 3, 5, 512) and idx.dtype == np.int32
    rec = oracle.dctq_inverse(idx, vol.shape, 1.0)
    assert np.abs(rec.astype(np.int32) - vol.astype(np.int32)).max() <= 1
    full = np.full((8, 8, 8), 1000, np.uint16)
    dc = oracle.dctq_forward(full, 1.0)[0, 0, 0]
    assert dc[0] == int(round(1000 * 512 ** 0.5)) and np.count_nonzero(dc[1:]) == 0
    # a 1-voxel volume is one replicated block: only the DC coefficient
    one = oracle.dctq_forward(np.array([[[77]]], np.uint16), 2.0)
    assert one.shape == (1, 1, 1, 512) and np.count_nonzero(one[0, 0, 0, 1:]) == 0
    coarse = oracle.dctq_inverse(oracle.dctq_forward(vol, 32.0), vol.shape, 32.0)
    assert 1.0 < np.abs(coarse.astype(np.float64) - vol).mean() < 32.0


def test_aggregation_denominator_is_a_convolution(oracle):
    """The identity behind the device path: den(v) = sum_b w_b win(v - c_b) equals
    the corner-weight volume C convolved with the separable window, here with the oracle's own
    window table and three causal 8-tap passes in float32."""
    rng = np.random.default_rng(5)
    shape = (24, 28, 33)
    _, win = oracle.tables()
    win = np.asarray(win, dtype=np.float32).reshape(8, 8, 8)
    k = (win[:, 0, 0] / win[0, 0, 0]).astype(np.float64)          # 1-D factor, k[0] = win[0,0,0]^(1/3)
    k *= float(win[0, 0, 0]) ** (1.0 / 3.0)
    np.testing.assert_allclose(np.einsum("i,j,k->ijk", k, k, k), win, rtol=2e-6)
    den = np.zeros(shape, np.float64)
    C = np.zeros(shape, np.float32)
    for _ in range(300):
        c = [int(rng.integers(0, n - 7)) for n in shape]
        w = np.float32(rng.uniform(1e-6, 1e-3))
        den[c[0]:c[0] + 8, c[1]:c[1] + 8, c[2]:c[2] + 8] += np.float64(w) * win
        C[c[0], c[1], c[2]] += w
    k32 = k.astype(np.float32)
    out = C.copy()
    for axis in (2, 1, 0):                                        # x, y, z: out(i) = sum_t k[t] in(i - t)
        acc = np.zeros_like(out)
        for t in range(8):
            sl_dst = [slice(None)] * 3
            sl_src = [slice(None)] * 3
            sl_dst[axis] = slice(t, None)
            sl_src[axis] = slice(0, out.shape[axis] - t)
            acc[tuple(sl_dst)] += k32[t] * out[tuple(sl_src)]
        out = acc
    np.testing.assert_allclose(out, den, rtol=2e-5, atol=1e-12)


def test_cpu_port_equals_the_oracle(oracle):
    """oracle/exabm4d_cpu_port.c (bench.py's cpu_baseline: shared cell sums, SIMD over dx and over
    transform lines, coloured parallel scatter) against the checker: match tables and 4-D
    transforms bit-identical and -- the aggregation sums being integers since round 4, DESIGN.md 3.8 -- the
    stage sums, the normalised estimates and the uint16 results too, whatever the order the port's
    threads add in -- on aligned and ragged extents, one thread and several."""
    rng = np.random.default_rng(7)
    for K in (1, 2, 4, 8, 16):
        g = rng.normal(0, 300, (K, 8, 8, 8)).astype(np.float32)
        spec = oracle.group_transform(g)
        np.testing.assert_array_equal(oracle.group_transform(g, port=True), spec)
        np.testing.assert_array_equal(oracle.group_transform(spec, inverse=True, port=True),
                                      oracle.group_transform(spec, inverse=True))
    for shape, threads in (((24, 28, 32), 3), ((26, 31, 21), 1), ((33, 24, 40), 4)):
        oracle.set_threads(threads)
        vol, _ = synth_volume(shape, seed=sum(shape))
        for c_match in (3.0, 0.6):
            keys = oracle.blockmatch(vol, 24.0, c_match)
            np.testing.assert_array_equal(oracle.blockmatch(vol, 24.0, c_match, port=True), keys)
        keys = oracle.blockmatch(vol, 24.0)
        E = oracle.data_exp(vol)
        NUM, CW = oracle.stage_q(vol, keys, 24.0, E)
        PN, PC = oracle.stage_q(vol, keys, 24.0, E, port=True)
        np.testing.assert_array_equal(PN, NUM)
        np.testing.assert_array_equal(PC, CW)
        num, den = oracle.stage(vol, keys, 24.0)
        pn, pd = oracle.stage(vol, keys, 24.0, port=True)
        np.testing.assert_array_equal(pn, num)
        np.testing.assert_array_equal(pd, den)
        basic = oracle.normalize(num, den)
        wn, wd = oracle.stage(vol, keys, 24.0, basic=basic)
        qn, qd = oracle.stage(vol, keys, 24.0, basic=basic, port=True)
        np.testing.assert_array_equal(qn, wn)
        np.testing.assert_array_equal(qd, wd)
    oracle.set_threads(len(os.sched_getaffinity(0)))
    u16, _ = synth_volume((40, 36, 44), seed=3, as_u16=True)
    a = oracle.bm4d_u16(u16, 24.0, 37.0)
    b = oracle.bm4d_u16(u16, 24.0, 37.0, port=True)
    np.testing.assert_array_equal(a, b)
    oracle.set_threads(1)                      # ... and the thread count does not matter either
    np.testing.assert_array_equal(oracle.bm4d_u16(u16, 24.0, 37.0, port=True), b)
    oracle.set_threads(len(os.sched_getaffinity(0)))


# ---- round 4: the order-independent aggregation ---------------------------------------------
def test_wiener_reciprocal_is_within_an_ulp_and_bit_defined(oracle):
    """R(d) (3.7): integer-subtraction seed + three fused Newton steps -- restated here with numpy's IEEE
    float32 / float64 operations, bit for bit, and within 1 ulp of 1/d over 60 binades."""
    rng = np.random.default_rng(11)
    d = np.exp2(rng.uniform(-30, 30, 4000)).astype(np.float32)
    d[:4] = np.float32([1.0, 576.0, 2.0 ** -20, 3.0e9])
    r = (np.uint32(0x7EF311C7) - d.view(np.uint32)).view(np.float32)
    def fma(a, b, c):
        # fma of float32 operands = float32(exact a*b + c): a*b is exact in float64 and the sum of a 48-bit
        # product and a 24-bit addend rounds once in float64 only beyond 2^-53 -- far below float32
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    one = np.ones_like(d)
    for _ in range(3):
        r = fma(fma(-d, r, one), r, r)         # Newton: r (2 - d r)
    got = np.array([oracle.rcp_nr(x) for x in d], dtype=np.float32)
    np.testing.assert_array_equal(got, r)
    exact = 1.0 / d.astype(np.float64)
    assert np.max(np.abs(got.astype(np.float64) - exact) / np.spacing(exact.astype(np.float32)).astype(np.float64)) < 1.0


def test_data_exponent_rule(oracle):
    """3.8: E = the exponent with max |v| < 2^E, read off the largest |v| bit pattern."""
    for v, E in ((1.0, 1), (0.99, 0), (65535.0, 16), (65536.0, 17), (-3000.0, 12), (1e-3, -9), (0.0, -126)):
        vol = np.zeros((8, 8, 8), np.float32)
        vol[3, 4, 5] = v
        assert oracle.data_exp(vol) == E, (v, E)
        assert abs(v) < 2.0 ** E


def test_integer_sums_follow_their_definition(oracle):
    """3.8 restated in float64 / Python integers for a handful of groups: NUM = sum of
    rint(est * fl32(u * win) * 2^(43 - E)) (half-to-even), CW = sum of rint(u * 2^40) on block corners, and
    the staged outputs num = fl32(fl64(NUM) 2^(E - 43)), den = the three 8-tap fp32 passes over fl32(CW 2^-40)."""
    vol, _ = synth_volume((16, 12, 20), seed=21)
    keys = oracle.blockmatch(vol, SIGMA, 3.0)
    E = oracle.data_exp(vol)
    NUM, CW = oracle.stage_q(vol, keys, SIGMA, E)
    _, win = oracle.tables()
    win = np.asarray(win, np.float32).reshape(8, 8, 8)
    thr = np.float32(np.float64(np.float32(2.7)) * np.float64(np.float32(SIGMA)))
    want_num = np.zeros(vol.shape, dtype=object)
    want_cw = np.zeros(vol.shape, dtype=object)
    want_num[...] = 0
    want_cw[...] = 0
    pz, py, px = (oracle.grid_positions(n) for n in vol.shape)
    for iz, rz in enumerate(pz):
        for iy, ry in enumerate(py):
            for ix, rx in enumerate(px):
                disp = [d for d, _ in oracle.decode_keys(keys[iz, iy, ix])]
                K = 1
                while K * 2 <= len(disp):
                    K *= 2
                disp = disp[:K]
                g = np.stack([vol[rz + a:rz + a + 8, ry + b:ry + b + 8, rx + c:rx + c + 8] for a, b, c in disp])
                spec = oracle.group_transform(g)
                keep = np.abs(spec) >= thr
                u = np.float32(1.0) / np.float32(max(int(keep.sum()), 1))
                est = oracle.group_transform(np.where(keep, spec, np.float32(0)), inverse=True)
                U = int(np.rint(np.float64(u) * 2.0 ** 40))
                ww = (u * win).astype(np.float32)
                terms = np.rint(est.astype(np.float64) * ww.astype(np.float64)[None] * 2.0 ** (43 - E))
                for j, (a, b, c) in enumerate(disp):
                    want_cw[rz + a, ry + b, rx + c] += U
                    sl = (slice(rz + a, rz + a + 8), slice(ry + b, ry + b + 8), slice(rx + c, rx + c + 8))
                    want_num[sl] += terms[j].astype(np.int64).astype(object)
    np.testing.assert_array_equal(NUM, want_num.astype(np.int64))
    np.testing.assert_array_equal(CW, want_cw.astype(np.int64))
    num, den = oracle.stage(vol, keys, SIGMA)
    np.testing.assert_array_equal(num, (NUM.astype(np.float64) * 2.0 ** (E - 43)).astype(np.float32))
    k = np.kaiser(8, 2.0).astype(np.float32)
    out = (CW.astype(np.float64) * 2.0 ** -40).astype(np.float32)
    for axis in (2, 1, 0):
        acc = np.zeros_like(out)
        for t in range(8):                     # acc = fma(k[t], in(i - t), acc), t ascending, from +0
            dst = [slice(None)] * 3
            src = [slice(None)] * 3
            dst[axis] = slice(t, None)
            src[axis] = slice(0, out.shape[axis] - t)
            acc[tuple(dst)] = (np.float64(k[t]) * out[tuple(src)].astype(np.float64) + acc[tuple(dst)].astype(np.float64)).astype(np.float32)
        out = acc
    np.testing.assert_array_equal(den, out)


# ---- the C oracle against an independent numpy reading of DESIGN.md section 3 (tests/bm4d_pyref.py) ----------
@pytest.mark.parametrize("shape,seed", [((12, 13, 16), 3), ((16, 16, 16), 4), ((9, 20, 24), 5), ((8, 8, 30), 6)])
def test_oracle_agrees_with_an_independent_restatement_of_the_specification(oracle, shape, seed):
    """BM4D is "parity unpinned" (the reference's bm4d wheel is absent); what CAN be pinned is that the
    oracle says what the specification text says.  tests/bm4d_pyref.py was written from DESIGN.md 3.1-3.8
    alone: grid, candidate set, fp32 fmaf-chain distances, packed keys and selection come out BIT-EXACT
    (both stages, ragged extents with clamped last grid points); the collaborative filtering -- there
    from the float64 definitions of DCT-II (x) Haar, hard threshold, Wiener weights, Kaiser window and
    aggregation -- to 2e-5 on the denominators and 1e-3 counts on the estimates."""
    import bm4d_pyref as R
    vol, _ = synth_volume(shape, seed=seed)
    keys = oracle.blockmatch(vol, SIGMA, 3.0)
    np.testing.assert_array_equal(R.blockmatch(vol, SIGMA, 3.0), keys)
    assert oracle.keymax(SIGMA, 3.0) == R.keymax(SIGMA, 3.0) and oracle.keymax(SIGMA, 0.6) == R.keymax(SIGMA, 0.6)
    assert oracle.grid_positions(shape[1]).tolist() == R.grid(shape[1])
    num_r, den_r = R.stage(vol, keys, SIGMA)
    num_o, den_o = oracle.stage(vol, keys, SIGMA)
    np.testing.assert_allclose(den_o, den_r, rtol=2e-5)
    assert np.max(np.abs(num_o / den_o - num_r / den_r)) < 1e-3
    basic = (num_o / den_o).astype(np.float32)
    kw = oracle.blockmatch(basic, SIGMA, 0.6)
    np.testing.assert_array_equal(R.blockmatch(basic, SIGMA, 0.6), kw)
    num_r, den_r = R.stage(vol, kw, SIGMA, basic=basic)
    num_o, den_o = oracle.stage(vol, kw, SIGMA, basic=basic)
    np.testing.assert_allclose(den_o, den_r, rtol=5e-5)
    assert np.max(np.abs(num_o / den_o - num_r / den_r)) < 1e-3


def test_oracle_pipeline_against_the_independent_restatement(oracle):
    """Whole two-stage pipeline, incl. integer-valued counts minus an offset (the uint16 form) and a
    constant volume (every candidate ties at distance 0)."""
    import bm4d_pyref as R
    vol = (synth_volume((12, 16, 16), seed=9, as_u16=True)[0].astype(np.float32) - np.float32(37.0))
    want = R.bm4d(vol, SIGMA)
    got = oracle.bm4d(vol, SIGMA)
    assert np.max(np.abs(got - want)) < 2e-2 and psnr(got, want, 1000.0) > 90.0
    flat = np.full((9, 9, 12), 5.0, dtype=np.float32)
    np.testing.assert_array_equal(R.blockmatch(flat, SIGMA, 3.0), oracle.blockmatch(flat, SIGMA, 3.0))
    # the hard-threshold stage keeps a constant exactly; the Wiener stage scales its DC by
    # W = Y^2 / (Y^2 + sigma^2) < 1 (3.7 filters the DC like any coefficient): both readings agree on that
    assert np.max(np.abs(oracle.bm4d(flat, SIGMA, stages=1) - 5.0)) < 1e-4
    assert np.max(np.abs(oracle.bm4d(flat, SIGMA) - R.bm4d(flat, SIGMA))) < 1e-4
    dc2 = (5.0 * np.sqrt(512.0 * 16.0)) ** 2
    assert abs(float(oracle.bm4d(flat, SIGMA)[4, 4, 6]) - 5.0 * dc2 / (dc2 + SIGMA ** 2)) < 1e-4


def test_match_tables_are_never_empty(oracle):
    """a block that holds an infinity or a NaN has no admissible distance, not even to itself,
    and gets the one-entry table [0]; every other table is what it was.  Oracle and CPU port agree, and the
    C pipelines run through such a volume without touching anything they should not (keys name blocks)."""
    vol, _ = synth_volume((16, 20, 24), seed=5)
    want = oracle.blockmatch(vol, 24.0, 3.0)
    assert (want[..., 0] == 0).all()
    bad = vol.copy()
    bad[5, 6, 7] = np.nan
    bad[12, 15, 20] = np.inf
    got = oracle.blockmatch(bad, 24.0, 3.0)
    np.testing.assert_array_equal(got, oracle.blockmatch(bad, 24.0, 3.0, port=True))
    assert (got[..., 0] == 0).all()
    pos = [oracle.grid_positions(n) for n in bad.shape]
    for iz, z in enumerate(pos[0]):
        for iy, y in enumerate(pos[1]):
            for ix, x in enumerate(pos[2]):
                if not np.isfinite(bad[z:z + 8, y:y + 8, x:x + 8]).all():
                    assert (got[iz, iy, ix, 1:] == 0xFFFFFFFF).all()
    with pytest.raises(ValueError, match="working range"):
        oracle.bm4d(bad, 24.0)
    out = oracle.bm4d(bad, 24.0, data_exp=oracle.data_exp(vol))        # past the wrapper: runs, garbage out
    assert out.shape == bad.shape
