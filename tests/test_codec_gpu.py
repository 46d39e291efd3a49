"""Row f-1 / BASELINE config 5 entropy coder on the GPU (csrc/rans_kernels.hip through the C-ABI):
chunk streams bit-identical to the oracle restatement of EXAC v1, exact decode round trips, the
committed format vectors, ragged chunk grids, int32 quantisation indices, compute_cratio with the
device codec, malformed streams."""
import os

import numpy as np
import pytest

from util import synth_volume

from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.utils import dct_quant as Q
from aind_exaspim_image_compression.utils import img_util
from aind_exaspim_image_compression.utils.chunk_codec import EncodedVolume, ShuffleRansCodec
from oracle import codec_oracle as co

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exac_v1.npz")


def denoised_like(shape, seed):
    rng = np.random.default_rng(seed)
    a = np.clip(rng.normal(37, 2.0, shape), 0, 65535)
    zz = np.arange(shape[0])[:, None, None]
    a = a + 900.0 * np.exp(-((zz - shape[0] / 2.0) ** 2) / 18.0) * (rng.random(shape) < 0.3)
    return np.rint(a).astype(np.uint16)


def check_against_oracle(enc, vol, chunk):
    want = [co.encode(c) for c in co.chunks(vol, chunk)]
    assert len(want) == len(enc.sizes)
    np.testing.assert_array_equal(enc.sizes, [len(w) for w in want])
    assert np.all(enc.offsets[:-1] % 16 == 0) and int(enc.offsets[-1]) == enc.data.size
    for i, w in enumerate(want):
        assert enc.chunk_bytes(i) == w, f"chunk {i} differs from the oracle"
        pad = enc.data[int(enc.offsets[i]) + len(w):int(enc.offsets[i + 1])]
        assert not pad.any()


def test_committed_format_vectors():
    g = np.load(GOLD)
    for name in sorted(k[:-3] for k in g.files if k.endswith("_in")):
        arr, want = g[name + "_in"], g[name + "_bytes"].tobytes()
        codec = ShuffleRansCodec(arr.dtype.itemsize)
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
])
def test_uint16_volumes_bit_identical_and_round_trip(shape, chunk):
    vol = denoised_like(shape, seed=sum(shape))
    vol.reshape(-1)[:: max(1, vol.size // 11)] = 65535
    codec = ShuffleRansCodec(2)
    enc = codec.encode_volume(vol, chunk)
    check_against_oracle(enc, vol, chunk)
    np.testing.assert_array_equal(codec.decode_volume(enc).reshape(shape), vol)
    np.testing.assert_array_equal(codec.chunk_sizes(vol, chunk), enc.sizes)


def test_noise_constant_and_tiny_chunks():
    rng = np.random.default_rng(2)
    codec = ShuffleRansCodec(2)
    for vol in (rng.integers(0, 65536, (64, 64, 64)).astype(np.uint16),          # incompressible
                np.full((64, 64, 70), 37, dtype=np.uint16),                        # constant planes
                np.array([[[513]]], dtype=np.uint16),
                (np.arange(64 * 64 * 64) % 251).astype(np.uint16).reshape(64, 64, 64)):
        enc = codec.encode_volume(vol, (64, 64, 64))
        check_against_oracle(enc, vol, (64, 64, 64))
        np.testing.assert_array_equal(codec.decode_volume(enc).reshape(vol.shape), vol)
        assert int(enc.offsets[-1]) <= _native.codec_volume_bound(2, vol.shape, (64, 64, 64))


def test_per_chunk_calls_equal_the_batched_call():
    """codec.encode(chunk) of compute_cratio's loop == the chunk's stream inside encode_volume."""
    vol = denoised_like((64, 100, 128), seed=4)
    codec = ShuffleRansCodec(2)
    enc = codec.encode_volume(vol)
    for i, c in enumerate(co.chunks(vol, (64, 64, 64))):
        b = codec.encode(c)
        assert b == enc.chunk_bytes(i)
        np.testing.assert_array_equal(codec.decode(b).reshape(c.shape), c)
    out = np.empty(64 * 64 * 64, dtype=np.uint16)
    assert codec.decode(enc.chunk_bytes(0), out=out) is out


def test_quantisation_indices_int32(oracle):
    """denoise-like volume -> DCT quantiser -> entropy coder (config 5's chain), indices coded as
    chunks of 2^18 consecutive values."""
    vol = synth_volume((40, 64, 72), seed=9, as_u16=True)[0]
    for q in (2.0, 24.0):
        idx = Q.quantise(vol, q)
        flat = idx.reshape(1, 1, -1)
        codec = ShuffleRansCodec(4)
        enc = codec.encode_volume(flat, chunk=(1, 1, 1 << 18))
        check_against_oracle(enc, flat, (1, 1, 1 << 18))
        back = codec.decode_volume(enc).reshape(idx.shape)
        np.testing.assert_array_equal(back, idx)
        assert enc.nbytes < idx.nbytes / 4
    ext = np.array([0, -1, 1, -2 ** 30, 2 ** 30, 255, -256, 65536] * 40, dtype=np.int32)
    assert ShuffleRansCodec(4).encode(ext) == co.encode(ext)


def test_compute_cratio_with_the_device_codec():
    """reference compute_cratio(img, codec, patch_shape) (utils/img_util.py:401-441): the batched
    path, the per-chunk loop with the same codec, and the oracle's sizes agree exactly."""
    vol = denoised_like((70, 128, 96), seed=6)
    codec = ShuffleRansCodec(2)
    got = img_util.compute_cratio(vol, codec)

    class Loop:                       # same codec without the batched entry: the reference's loop
        def encode(self, chunk):
            return codec.encode(chunk)

    packed = sum(len(co.encode(c)) for c in co.chunks(vol, (64, 64, 64)))
    assert got == img_util.compute_cratio(vol, Loop()) == round(vol.nbytes / packed, 2)
    assert got == img_util.compute_cratio(vol[None, None], codec)       # 5-D input like the reference
    assert got <= img_util.shuffled_entropy_cratio(vol)                  # never above the order-0 floor
    assert got > 0.97 * img_util.shuffled_entropy_cratio(vol) and got > 3.5


def test_errors():
    codec = ShuffleRansCodec(2)
    with pytest.raises(ValueError):
        ShuffleRansCodec(3)
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
