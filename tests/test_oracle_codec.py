"""CPU: the chunk entropy coder's restatement (oracle/exac_codec.c; EXAC v1, DESIGN.md 3.11, and
EXAC v2) -- exact round trips, the committed format vectors of both versions, an
independent pure-Python reading of the v2 format, the normalisation rules, the reciprocal identity
the HIP kernels rely on, sizes against entropy floors, malformed streams."""
import os

import numpy as np
import pytest

from oracle import codec_oracle as co

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exac_v1.npz")
GOLD2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exac_v2.npz")


@pytest.mark.parametrize("version,path,count", [(1, GOLD, 7), (2, GOLD2, 13)])
def test_committed_format_vectors(version, path, count):
    g = np.load(path)
    names = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    assert len(names) == count
    for name in names:
        arr, want = g[name + "_in"], g[name + "_bytes"].tobytes()
        assert co.encode(arr, version=version) == want, name
        back, used = co.decode(want, arr.size, arr.dtype.itemsize)
        assert used == len(want)
        np.testing.assert_array_equal(back, arr.reshape(-1))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 20000])
@pytest.mark.parametrize("dtype", [np.uint16, np.int32])
def test_round_trip_ragged_lengths(n, dtype):
    rng = np.random.default_rng(n)
    if dtype == np.uint16:
        a = np.clip(rng.normal(200, 90, n), 0, 65535).round().astype(dtype)
        a[:: 17] = rng.integers(0, 65536, len(a[:: 17]))
    else:
        a = rng.normal(0, 3, n).round().astype(dtype)
        a[:: 29] = rng.integers(-2 ** 30, 2 ** 30, len(a[:: 29]))
    for version in (1, 2):
        b = co.encode(a, version=version)
        assert len(b) <= (co.bound if version == 1 else co.bound2)(n, a.dtype.itemsize)
        back, used = co.decode(b, n, a.dtype.itemsize)
        assert used == len(b)
        np.testing.assert_array_equal(back, a)


def test_header_layout_and_constant_planes():
    a = np.full(1000, 0x0125, dtype=np.uint16)
    b = co.encode(a, version=1)
    assert b[:4] == b"EX\x01\x02" and int.from_bytes(b[4:8], "little") == 1000
    assert int.from_bytes(b[8:12], "little") == 0 and int.from_bytes(b[12:16], "little") == 0
    # two tables of bitmap + one frequency (4096), no words
    assert len(b) == 16 + 2 * (32 + 2)
    assert b[16 + 0x25 // 8] == 1 << (0x25 % 8) and int.from_bytes(b[48:50], "little") == 4096
    a[5] = 0x0126                                            # low plane now has two symbols
    b = co.encode(a, version=1)
    nw0 = int.from_bytes(b[8:12], "little")
    assert nw0 >= 128 and int.from_bytes(b[12:16], "little") == 0
    assert len(b) == 16 + (32 + 4) + (32 + 2) + 2 * nw0


def test_normalisation_rule():
    rng = np.random.default_rng(3)
    for trial in range(200):
        k = int(rng.integers(1, 257))
        cnt = np.zeros(256, dtype=np.uint32)
        sym = rng.choice(256, size=k, replace=False)
        cnt[sym] = np.maximum(1, (rng.pareto(0.6, k) * 3).astype(np.uint32))
        f = co.normalize(cnt).astype(np.int64)
        assert f.sum() == 4096 and np.all((f > 0) == (cnt > 0))
    # 255 rare symbols next to one dominant one: every rare symbol keeps a slot
    cnt = np.ones(256, dtype=np.uint32)
    cnt[7] = 10 ** 6
    f = co.normalize(cnt)
    assert f.sum() == 4096 and f[7] == 4096 - 255 and np.all(np.delete(f, 7) == 1)
    # ties go to the lowest symbol
    cnt = np.zeros(256, dtype=np.uint32)
    cnt[[3, 9, 200]] = 5
    f = co.normalize(cnt)
    assert f[3] == 1366 and f[9] == 1365 and f[200] == 1365
    assert co.normalize(np.zeros(256, np.uint32)).sum() == 0


def test_reciprocal_division_is_exact():
    """q = mulhi(x, rcp) >> shift == x / f for every frequency and the critical x < 2^31."""
    rng = np.random.default_rng(5)
    for f in range(1, 4097):
        top = (1 << 31) - 1
        xs = {0, 1, f - 1, f, f + 1, top, top - f, (top // f) * f, (top // f) * f - 1,
              (f << 19) - 1, 1 << 15, (1 << 16) - 1}
        xs.update(int(v) for v in rng.integers(0, 1 << 31, 6))
        for x in xs:
            assert co.check_reciprocal(min(max(x, 0), top), f), (x, f)


def test_size_against_entropy_floor():
    """A 64^3 chunk of denoised-like counts: never below the order-0 entropy of its byte planes,
    within 1 % + tables + final states of it."""
    rng = np.random.default_rng(11)
    for spread in (1.5, 6.0, 40.0):
        a = np.clip(rng.normal(37, spread, (64, 64, 64)), 0, 65535).round().astype(np.uint16)
        a[20:24, 10:50, 30:34] += 2000
        size, floor = len(co.encode(a, version=1)), co.plane_entropy_bytes(a)
        assert floor <= size <= 1.01 * floor + 16 + 2 * (32 + 512 + 256)
    raw = rng.integers(0, 65536, (64, 64, 64)).astype(np.uint16)   # incompressible
    assert len(co.encode(raw, version=1)) <= co.bound(raw.size, 2)
    assert len(co.encode(raw, version=1)) < 1.005 * raw.nbytes


def test_malformed_streams_are_rejected():
    a = np.arange(500, dtype=np.uint16)
    b = bytearray(co.encode(a, version=1))
    for cut in (0, 7, 15, 40, len(b) - 2):
        with pytest.raises(ValueError):
            co.decode(bytes(b[:cut]), a.size, 2)
    with pytest.raises(ValueError):
        co.decode(bytes(b), a.size + 1, 2)                    # element count mismatch
    with pytest.raises(ValueError):
        co.decode(bytes(b), a.size, 4)                        # typesize mismatch
    bad = bytearray(b)
    bad[2] = 9                                                 # unknown version
    with pytest.raises(ValueError):
        co.decode(bytes(bad), a.size, 2)
    bad = bytearray(b)
    bad[48] ^= 0xFF                                            # a frequency: the sum is no longer 4096
    with pytest.raises(ValueError):
        co.decode(bytes(bad), a.size, 2)


# ---- EXAC v2 --------------------------------------------------------------------------------------
def _model_v2(a):
    """numpy restatement of the v2 MODEL (taps, prediction, contexts, symbols) for chunks whose rows
    are x-rows (ex a multiple of 64): (sym, raw bits, ctx) per element."""
    a = np.asarray(a)
    ez, ey, ex = a.shape
    assert ex % 64 == 0 and ey * ex <= 8000
    c = a.astype(np.int64)
    U = np.zeros(a.shape, bool)
    U[:, 1:] = True
    B = np.zeros(a.shape, bool)
    B[1:] = True
    up = np.zeros_like(c)
    up[:, 1:] = c[:, :-1]
    bk = np.zeros_like(c)
    bk[1:] = c[:-1]
    if a.dtype == np.uint16:
        pred = np.where(U & B, (up + bk + 1) >> 1, np.where(U, up, np.where(B, bk, 0)))
        r = ((c - pred + 32768) % 65536) - 32768
    else:
        r = c
    u = np.where(r >= 0, 2 * r, -2 * r - 1)
    mag = np.minimum((u + 1) >> 1, 127)
    mu = np.zeros_like(mag)
    mu[:, 1:] = mag[:, :-1]
    mb = np.zeros_like(mag)
    mb[1:] = mag[:-1]
    act = np.where(U & B, mu + mb, np.where(U, 2 * mu, np.where(B, 2 * mb, 0)))
    ctx = np.searchsorted(np.array([1, 2, 3, 4, 5, 6, 8, 10, 13, 17, 22, 30, 45, 70, 120]), act, side="right")
    w = np.maximum(u - 32, 0)
    cat = np.floor(np.log2((w >> 2) + 1)).astype(np.int64)
    sym = np.where(u < 32, u, 32 + cat)
    nb = np.where(u < 32, 0, 2 + cat)
    return sym, nb, ctx


def _ideal_bytes_v2(a):
    """what the v2 model costs with ideal arithmetic coding of the normalised tables"""
    sym, nb, ctx = _model_v2(a)
    bits = float(nb.sum())
    for q in range(16):
        s = sym[ctx == q]
        if s.size:
            cnt = np.bincount(s, minlength=64)
            f = co.normalize2(cnt).astype(np.float64)
            bits += float(-(cnt[cnt > 0] * np.log2(f[cnt > 0] / 4096.0)).sum())
    return bits / 8.0


def test_v2_independent_python_decoder_reads_the_committed_vectors():
    """tests/exac2_pyref.py was written from the format text, not from the C file."""
    import exac2_pyref
    g = np.load(GOLD2)
    for name in sorted(k[:-3] for k in g.files if k.endswith("_in")):
        arr = g[name + "_in"]
        got, shape = exac2_pyref.decode(g[name + "_bytes"].tobytes())
        assert shape == co.shape3(arr.shape), name
        np.testing.assert_array_equal(got, arr.reshape(-1), err_msg=name)


@pytest.mark.parametrize("shape", [(5, 7, 3), (1, 1, 1), (1, 1, 200), (3, 100, 12), (2, 3, 64), (17, 64, 64),
                                   (4, 8, 65), (1, 130, 70), (3, 2, 9000), (2, 130, 64), (70, 1, 1)])
def test_v2_round_trip_of_3d_shapes(shape):
    """narrow rows (taps several x-rows up), planes smaller than a row of 64, rows and planes beyond
    the tap limit (8000 elements: the tap is not used), both element kinds, extreme values."""
    rng = np.random.default_rng(sum(shape))
    a = np.clip(rng.normal(300, 40, shape), 0, 65535).round().astype(np.uint16)
    a.reshape(-1)[::13] = rng.integers(0, 65536, a.reshape(-1)[::13].size)
    i = rng.laplace(0, 9, shape).round().astype(np.int32)
    i.reshape(-1)[0], i.reshape(-1)[-1] = -2 ** 31, 2 ** 31 - 1
    for arr in (a, i, np.zeros(shape, np.uint16), np.full(shape, 65535, np.uint16)):
        b = co.encode(arr)
        assert b[:3] == b"EX\x02" and len(b) <= co.bound2(arr.size, arr.dtype.itemsize)
        back, used = co.decode(b, arr.size, arr.dtype.itemsize)
        assert used == len(b)
        np.testing.assert_array_equal(back.reshape(shape), arr)


def test_v2_header_and_silent_chunks():
    z = np.zeros((4, 8, 64), np.uint16)
    b = co.encode(z)
    assert len(b) == 276 + 2                                  # one symbol in one context, no words
    assert [int.from_bytes(b[4 + 4 * k:8 + 4 * k], "little") for k in range(4)] == [z.size, 8, 64, 0]
    assert int.from_bytes(b[20:28], "little") == 1 and b[276] == 255 and b[277] == 15   # F - 1 = 4095
    z[2, 3, 4] = 9
    b = co.encode(z)
    assert int.from_bytes(b[16:20], "little") >= 128          # now there is something to code


def test_v2_normalisation_rule():
    rng = np.random.default_rng(4)
    for trial in range(300):
        k = int(rng.integers(1, 65))
        cnt = np.zeros(64, dtype=np.uint32)
        cnt[rng.choice(64, size=k, replace=False)] = np.maximum(1, (rng.pareto(0.5, k) * 3).astype(np.uint32))
        f = co.normalize2(cnt).astype(np.int64)
        assert f.sum() == 4096 and np.all((f > 0) == (cnt > 0))
    cnt = np.ones(64, dtype=np.uint32)
    cnt[7] = 10 ** 6
    f = co.normalize2(cnt)
    assert f[7] == 4096 - 63 and np.all(np.delete(f, 7) == 1)
    cnt = np.zeros(64, dtype=np.uint32)
    cnt[[3, 9, 20]] = 5                                        # floor 1365 each: the deficit of 1 to the lowest
    assert list(co.normalize2(cnt)[[3, 9, 20]]) == [1366, 1365, 1365]
    # an excess (44 rare symbols forced up to 1) comes off the largest F, lowest symbol on ties, at once
    cnt = np.full(64, 1, dtype=np.uint32)
    cnt[:20] = 10 ** 5
    f = co.normalize2(cnt).astype(np.int64)
    assert f.sum() == 4096 and f[0] == 204 - 28 and np.all(f[1:20] == 204) and np.all(f[20:] == 1)
    assert co.normalize2(np.zeros(64, np.uint32)).sum() == 0


def test_v2_size_against_its_model_and_against_v1():
    """The coded size is the model's ideal cost (numpy restatement of taps / contexts / symbols) to
    within 0.5 % + header + tables + final states, and well below v1 on structured data."""
    rng = np.random.default_rng(12)
    zz, yy, xx = np.meshgrid(np.arange(64), np.arange(64), np.arange(64), indexing="ij")
    ridge = 2500.0 * np.exp(-((yy - 30.0) ** 2 + (xx - 20.0 - 0.2 * zz) ** 2) / 30.0)
    for spread in (1.5, 6.0):
        a = np.clip(37 + ridge + rng.normal(0, spread, ridge.shape), 0, 65535).round().astype(np.uint16)
        size, ideal = len(co.encode(a)), _ideal_bytes_v2(a)
        assert ideal <= size <= 1.005 * ideal + 276 + 16 * 128 + 256 + 64, (size, ideal)
        assert size < 0.9 * len(co.encode(a, version=1))
    idx = rng.laplace(0, 2.0, (512, 8, 64)).round().astype(np.int32)
    idx[:, 0, 0] += rng.integers(-3000, 3000, 512)
    size, ideal = len(co.encode(idx)), _ideal_bytes_v2(idx)
    assert ideal <= size <= 1.005 * ideal + 276 + 16 * 128 + 256 + 64
    raw = rng.integers(0, 65536, (64, 64, 64)).astype(np.uint16)   # incompressible: bounded expansion
    assert len(co.encode(raw)) < 1.02 * raw.nbytes


def test_v2_malformed_streams_are_rejected():
    a = np.clip(np.random.default_rng(1).normal(100, 30, (3, 10, 64)), 0, 65535).astype(np.uint16)
    b = bytearray(co.encode(a))
    for cut in (0, 7, 19, 275, 300, len(b) - 2):
        with pytest.raises(ValueError):
            co.decode(bytes(b[:cut]), a.size, 2)
    with pytest.raises(ValueError):
        co.decode(bytes(b), a.size + 64, 2)
    with pytest.raises(ValueError):
        co.decode(bytes(b), a.size, 4)
    used = next(p for p in range(20, 148) if b[p])            # a byte of a `present` bitmap with symbols in it
    for pos, val in ((8, 7), (12, 0), (used, 0), (155, 0xFF), (276, b[276] ^ 0xFF)):   # ey, ex, present, wide, a frequency
        bad = bytearray(b)
        bad[pos] = val
        with pytest.raises(ValueError):
            co.decode(bytes(bad), a.size, 2)
    bad = bytearray(b)
    bad[16:20] = (1 << 30).to_bytes(4, "little")               # more words than the stream holds
    with pytest.raises(ValueError):
        co.decode(bytes(bad), a.size, 2)


def test_v2_fuzz_c_encoder_against_the_independent_python_decoder():
    """120 random small chunks -- shapes with rows narrower and wider than a wave, planes smaller than
    a row, both element kinds, flat / noisy / spiky / extreme data -- coded by the C restatement and
    read back by tests/exac2_pyref.py (written from the format text alone), and by the C decoder."""
    import exac2_pyref
    rng = np.random.default_rng(20261006)
    for it in range(120):
        shape = tuple(int(v) for v in rng.integers(1, [7, 12, 90]))
        kind = int(rng.integers(0, 5))
        if rng.random() < 0.5:
            base = rng.normal(rng.choice([0, 37, 3000, 60000]), rng.choice([0.0, 1.0, 4.0, 40.0, 900.0]), shape)
            a = np.clip(base, 0, 65535).round().astype(np.uint16)
            if kind == 0:
                a.reshape(-1)[:: int(rng.integers(2, 9))] = rng.integers(0, 65536)
        else:
            a = rng.laplace(0, rng.choice([0.3, 2.0, 50.0, 1e5]), shape).round().astype(np.int64)
            a = np.clip(a, -2 ** 31, 2 ** 31 - 1).astype(np.int32)
            if kind == 1:
                a.reshape(-1)[0] = -2 ** 31
        b = co.encode(a)
        got, shp = exac2_pyref.decode(b)
        assert shp == co.shape3(shape), (it, shape)
        np.testing.assert_array_equal(got, a.reshape(-1), err_msg=f"iteration {it}, shape {shape}")
        back, used = co.decode(b, a.size, a.dtype.itemsize)
        assert used == len(b)
        np.testing.assert_array_equal(back, a.reshape(-1))
