"""Row f-1 / BASELINE config 5 entropy coder on the GPU (csrc/rans2_kernels.hip: EXAC v2, the default;
csrc/rans_kernels.hip: EXAC v1 -- both through the C-ABI): chunk streams bit-identical to the oracle
restatement, exact decode round trips, the committed format vectors of both versions, ragged chunk
grids, int32 quantisation indices, compute_cratio with the device codec, malformed streams and
containers."""
import os

import numpy as np
import pytest

from util import synth_volume

from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.utils import dct_quant as Q
from aind_exaspim_image_compression.utils import img_util
from aind_exaspim_image_compression.utils.chunk_codec import EncodedVolume, ExacCodec, ShuffleRansCodec
from oracle import codec_oracle as co

pytestmark = pytest.mark.gpu
GOLD = {v: os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exac_v%d.npz" % v) for v in (1, 2)}
VERSIONS = pytest.mark.parametrize("version", [2, 1])


def denoised_like(shape, seed):
    rng = np.random.default_rng(seed)
    a = np.clip(rng.normal(37, 2.0, shape), 0, 65535)
    zz = np.arange(shape[0])[:, None, None]
    a = a + 900.0 * np.exp(-((zz - shape[0] / 2.0) ** 2) / 18.0) * (rng.random(shape) < 0.3)
    return np.rint(a).astype(np.uint16)


def check_against_oracle(enc, vol, chunk, version):
    want = [co.encode(c, version=version) for c in co.chunks(vol, chunk)]
    assert len(want) == len(enc.sizes)
    np.testing.assert_array_equal(enc.sizes, [len(w) for w in want])
    assert np.all(enc.offsets[:-1] % 16 == 0) and int(enc.offsets[-1]) == enc.data.size
    for i, w in enumerate(want):
        assert enc.chunk_bytes(i) == w, f"chunk {i} differs from the oracle"
        pad = enc.data[int(enc.offsets[i]) + len(w):int(enc.offsets[i + 1])]
        assert not pad.any()


@VERSIONS
def test_committed_format_vectors(version):
    g = np.load(GOLD[version])
    for name in sorted(k[:-3] for k in g.files if k.endswith("_in")):
        arr, want = g[name + "_in"], g[name + "_bytes"].tobytes()
        codec = ExacCodec(arr.dtype.itemsize, version=version)
        assert codec.encode(arr) == want, name
        np.testing.assert_array_equal(codec.decode(want), arr.reshape(-1))


@pytest.mark.parametrize("shape,chunk", [
    ((64, 64, 64), (64, 64, 64)),          # the reference's chunk (utils/img_util.py:401)
    ((70, 65, 130), (64, 64, 64)),         # ragged grid: truncated edge chunks on every axis
    ((40, 48, 96), (32, 32, 32)),
    ((3, 5, 1000), (2, 5, 300)),           # rows that straddle chunk rows (generic addressing)
    ((1, 1, 5000), (1, 1, 4096)),
    ((128, 64, 64), (64, 64, 64)),
    ((6, 10, 320), (4, 6, 128)),           # x extent a multiple of 64: row-cursor addressing
    ((9, 40, 12), (9, 40, 12)),            # rows narrower than a wave: v2 taps several x-rows up
    ((50, 3, 5), (50, 3, 5)),              # planes smaller than a row of 64
    ((2, 3, 9000), (2, 3, 9000)),          # rows beyond v2's tap limit
    ((2, 130, 64), (2, 130, 64)),          # planes beyond v2's tap limit
])
@VERSIONS
def test_uint16_volumes_bit_identical_and_round_trip(shape, chunk, version):
    vol = denoised_like(shape, seed=sum(shape))
    vol.reshape(-1)[:: max(1, vol.size // 11)] = 65535
    codec = ExacCodec(2, version=version)
    enc = codec.encode_volume(vol, chunk)
    check_against_oracle(enc, vol, chunk, version)
    np.testing.assert_array_equal(codec.decode_volume(enc).reshape(shape), vol)
    np.testing.assert_array_equal(codec.chunk_sizes(vol, chunk), enc.sizes)


@VERSIONS
def test_noise_constant_and_tiny_chunks(version):
    rng = np.random.default_rng(2)
    codec = ExacCodec(2, version=version)
    for vol in (rng.integers(0, 65536, (64, 64, 64)).astype(np.uint16),          # incompressible
                np.full((64, 64, 70), 37, dtype=np.uint16),                        # constant planes
                np.array([[[513]]], dtype=np.uint16),
                (np.arange(64 * 64 * 64) % 251).astype(np.uint16).reshape(64, 64, 64)):
        enc = codec.encode_volume(vol, (64, 64, 64))
        check_against_oracle(enc, vol, (64, 64, 64), version)
        np.testing.assert_array_equal(codec.decode_volume(enc).reshape(vol.shape), vol)
        assert int(enc.offsets[-1]) <= _native.codec_volume_bound(2, vol.shape, (64, 64, 64))


@VERSIONS
def test_per_chunk_calls_equal_the_batched_call(version):
    """codec.encode(chunk) of compute_cratio's loop == the chunk's stream inside encode_volume."""
    vol = denoised_like((64, 100, 128), seed=4)
    codec = ExacCodec(2, version=version)
    enc = codec.encode_volume(vol)
    for i, c in enumerate(co.chunks(vol, (64, 64, 64))):
        b = codec.encode(c)
        assert b == enc.chunk_bytes(i)
        np.testing.assert_array_equal(codec.decode(b).reshape(c.shape), c)
    out = np.empty(64 * 64 * 64, dtype=np.uint16)
    assert codec.decode(enc.chunk_bytes(0), out=out) is out


@VERSIONS
def test_quantisation_indices_int32(oracle, version):
    """denoise-like volume -> DCT quantiser -> entropy coder (config 5's chain), indices coded as
    chunks of 2^18 consecutive values: flat (v1's view) and as (512 blocks, 8, 64) -- the shape
    bench.py codes them in, where v2's "up" tap is the neighbouring frequency and its "back" tap
    the same coefficient of the block before."""
    vol = synth_volume((40, 64, 72), seed=9, as_u16=True)[0]
    codec = ExacCodec(4, version=version)
    for q in (2.0, 24.0):
        idx = Q.quantise(vol, q)
        for view, chunk in ((idx.reshape(1, 1, -1), (1, 1, 1 << 18)), (idx.reshape(-1, 8, 64), (512, 8, 64))):
            enc = codec.encode_volume(view, chunk=chunk)
            check_against_oracle(enc, view, chunk, version)
            back = codec.decode_volume(enc).reshape(idx.shape)
            np.testing.assert_array_equal(back, idx)
            assert enc.nbytes < idx.nbytes / 4
    ext = np.array([0, -1, 1, -2 ** 30, 2 ** 30, 255, -256, 65536, -2 ** 31, 2 ** 31 - 1] * 40, dtype=np.int32)
    assert codec.encode(ext) == co.encode(ext, version=version)
    np.testing.assert_array_equal(codec.decode(codec.encode(ext)), ext)


@VERSIONS
def test_compute_cratio_with_the_device_codec(version):
    """reference compute_cratio(img, codec, patch_shape) (utils/img_util.py:401-441): the batched
    path, the per-chunk loop with the same codec, and the oracle's sizes agree exactly."""
    vol = denoised_like((70, 128, 96), seed=6)
    codec = ExacCodec(2, version=version)
    got = img_util.compute_cratio(vol, codec)

    class Loop:                       # same codec without the batched entry: the reference's loop
        def encode(self, chunk):
            return codec.encode(chunk)

    packed = sum(len(co.encode(c, version=version)) for c in co.chunks(vol, (64, 64, 64)))
    assert got == img_util.compute_cratio(vol, Loop()) == round(vol.nbytes / packed, 2)
    assert got == img_util.compute_cratio(vol[None, None], codec)       # 5-D input like the reference
    if version == 1:
        assert got <= img_util.shuffled_entropy_cratio(vol)              # never above the order-0 floor
        assert got > 0.97 * img_util.shuffled_entropy_cratio(vol) and got > 3.5
    else:
        assert got > 3.0            # iid noise + sparse spikes: little for the predictor to use


def test_round2_name_is_the_v1_codec():
    a = denoised_like((8, 16, 64), seed=1)
    assert ShuffleRansCodec(2).encode(a) == co.encode(a, version=1) == ExacCodec(2, version=1).encode(a)
    np.testing.assert_array_equal(ExacCodec(2).decode(ShuffleRansCodec(2).encode(a)), a.reshape(-1))   # v2 object, v1 bytes


def test_v2_malformed_streams_and_containers():
    """Corrupt headers / tables / words and corrupt containers (offsets that are not
    ascending or point outside the data) raise ValueError -- the kernel never reads outside the
    buffer it was given."""
    codec = ExacCodec(2)
    a = denoised_like((3, 10, 64), seed=5)
    good = codec.encode(a)
    np.testing.assert_array_equal(codec.decode(good).reshape(a.shape), a)
    used = next(p for p in range(20, 148) if good[p])          # a byte of a `present` bitmap with symbols in it
    for pos, val in ((2, 9), (3, 4), (8, 7), (used, 0), (155, 0xFF), (276, good[276] ^ 0xFF)):
        bad = bytearray(good)                                   # version, typesize, ey, present, wide, a frequency
        bad[pos] = val
        with pytest.raises(ValueError):
            codec.decode(bytes(bad))
    for cut in (10, 275, 300, len(good) - 2):
        with pytest.raises(ValueError):
            codec.decode(good[:cut])
    bad = bytearray(good)
    bad[16:20] = (1 << 30).to_bytes(4, "little")
    with pytest.raises(ValueError):
        codec.decode(bytes(bad))
    vol = denoised_like((64, 64, 128), seed=6)
    for version in (2, 1):
        c = ExacCodec(2, version=version)
        enc = c.encode_volume(vol)
        for offsets in (np.array([0, 1 << 40, enc.data.size], np.uint64),             # far outside the data
                        np.array([enc.offsets[1], enc.offsets[0], enc.offsets[2]], np.uint64),   # descending
                        np.array([0, enc.offsets[1] + 1, enc.offsets[2]], np.uint64)):  # odd
            with pytest.raises(ValueError):
                c.decode_volume(EncodedVolume(enc.data, offsets, enc.sizes, enc.shape, enc.chunk, 2))
        # the device check itself (the Python wrapper's is bypassed): offsets beyond in_bytes
        ctx = _native.context(0)
        d_in, d_vol = ctx.to_device(enc.data), ctx.alloc(vol.nbytes)
        d_off = ctx.to_device(np.array([0, enc.offsets[1], 1 << 33], np.uint64))
        with pytest.raises(ValueError):
            ctx.codec_decode(d_in, enc.data.size, d_off, 2, vol.shape, (64, 64, 64), d_vol)
        for buf in (d_in, d_vol, d_off):
            buf.free()


def test_errors():
    codec = ShuffleRansCodec(2)
    with pytest.raises(ValueError):
        ShuffleRansCodec(3)
    with pytest.raises(ValueError):
        ExacCodec(2, version=3)
    with pytest.raises(ValueError):
        codec.encode(np.zeros(8, dtype=np.float32))
    with pytest.raises(ValueError):
        codec.encode(np.zeros(0, dtype=np.uint16))
    with pytest.raises(ValueError):
        codec.decode(b"not a stream")
    good = bytearray(codec.encode(np.arange(300, dtype=np.uint16)))
    bad = bytearray(good)
    bad[48] ^= 0xFF                                                      # frequencies no longer sum to 4096
    with pytest.raises(ValueError):
        codec.decode(bytes(bad))
    with pytest.raises(ValueError):
        codec.decode(bytes(good[:60]))                                   # truncated
    enc = EncodedVolume(np.zeros(64, np.uint8), np.array([0, 64], np.uint64),
                        np.array([64], np.uint32), (1, 1, 10), (1, 1, 10), 2)
    with pytest.raises(ValueError):
        codec.decode_volume(enc)                                         # no magic
    np.testing.assert_array_equal(codec.decode(bytes(good)), np.arange(300, dtype=np.uint16))


def test_chunk_store_round_trip(tmp_path):
    """Round 4: write_zarr / read_zarr (utils/chunk_store.py) -- the coded volume as a local Zarr-v3-style
    directory in the reference's chunk layout (write_zarr, utils/img_util.py:898-950): 256^3 in 64^3 chunks and
    a ragged volume whose edge chunks are stored truncated; the files are codec.encode(chunk), their sizes sum
    to compute_cratio's denominator, a single chunk decodes by its key."""
    import json
    import os
    from aind_exaspim_image_compression.utils import chunk_store as S
    from aind_exaspim_image_compression.utils import img_util
    from aind_exaspim_image_compression.utils.chunk_codec import ExacCodec
    import bench
    vol = bench.synth_u16((256, 256, 256), seed=4)
    path = str(tmp_path / "den.zarr")
    ratio = S.write_zarr(vol, path)
    meta = json.load(open(os.path.join(path, "zarr.json")))
    assert meta["shape"] == [1, 1, 256, 256, 256] and meta["chunk_grid"]["configuration"]["chunk_shape"] == [1, 1, 64, 64, 64]
    stored = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(os.path.join(path, "c")) for f in fs)
    assert abs(ratio - vol.nbytes / stored) < 1e-12 and round(ratio, 2) == img_util.compute_cratio(vol, ExacCodec(2))
    back = S.read_zarr(path)
    assert back.shape == (1, 1, 256, 256, 256) and back.dtype == np.uint16
    np.testing.assert_array_equal(back[0, 0], vol)
    np.testing.assert_array_equal(S.read_chunk(path, 3, 0, 2), vol[192:256, 0:64, 128:192])
    assert open(os.path.join(path, S.chunk_key(1, 2, 3)), "rb").read() == ExacCodec(2).encode(vol[64:128, 128:192, 192:256])
    rag = bench.synth_u16((70, 65, 100), seed=5)
    p2 = str(tmp_path / "rag.zarr")
    S.write_zarr(rag[np.newaxis], p2)                       # 4-D in, promoted like the reference
    np.testing.assert_array_equal(S.read_zarr(p2)[0, 0], rag)
    np.testing.assert_array_equal(S.read_chunk(p2, 1, 1, 1), rag[64:70, 64:65, 64:100])     # truncated edge chunk
    with pytest.raises(IndexError):
        S.read_chunk(p2, 2, 0, 0)
    with pytest.raises(ValueError):
        S.write_zarr(np.zeros((2, 1, 8, 8, 8), np.uint16), p2)
