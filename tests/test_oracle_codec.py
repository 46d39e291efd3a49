"""CPU: the chunk entropy coder's restatement (oracle/exac_codec.c, EXAC v1) --
exact round trips, the committed format vectors, the normalisation rule, the reciprocal identity
the HIP kernels rely on, the size against the order-0 entropy floor, malformed streams."""
import os

import numpy as np
import pytest

from oracle import codec_oracle as co

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exac_v1.npz")


def test_committed_format_vectors():
    g = np.load(GOLD)
    names = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    assert len(names) == 7
    for name in names:
        arr, want = g[name + "_in"], g[name + "_bytes"].tobytes()
        assert co.encode(arr) == want, name
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
    b = co.encode(a)
    assert len(b) <= co.bound(n, a.dtype.itemsize)
    back, used = co.decode(b, n, a.dtype.itemsize)
    assert used == len(b)
    np.testing.assert_array_equal(back, a)


def test_header_layout_and_constant_planes():
    a = np.full(1000, 0x0125, dtype=np.uint16)
    b = co.encode(a)
    assert b[:4] == b"EX\x01\x02" and int.from_bytes(b[4:8], "little") == 1000
    assert int.from_bytes(b[8:12], "little") == 0 and int.from_bytes(b[12:16], "little") == 0
    # two tables of bitmap + one frequency (4096), no words
    assert len(b) == 16 + 2 * (32 + 2)
    assert b[16 + 0x25 // 8] == 1 << (0x25 % 8) and int.from_bytes(b[48:50], "little") == 4096
    a[5] = 0x0126                                            # low plane now has two symbols
    b = co.encode(a)
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
        size, floor = len(co.encode(a)), co.plane_entropy_bytes(a)
        assert floor <= size <= 1.01 * floor + 16 + 2 * (32 + 512 + 256)
    raw = rng.integers(0, 65536, (64, 64, 64)).astype(np.uint16)   # incompressible
    assert len(co.encode(raw)) <= co.bound(raw.size, 2)
    assert len(co.encode(raw)) < 1.005 * raw.nbytes


def test_malformed_streams_are_rejected():
    a = np.arange(500, dtype=np.uint16)
    b = bytearray(co.encode(a))
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
