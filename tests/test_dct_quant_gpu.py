"""Row f-1 transform quantiser on the GPU: quantisation indices bit-exact against
the oracle (integer work), reconstruction identical, edge replication for extents that are not
multiples of 8, and size-independent properties on a larger volume."""
import numpy as np
import pytest

from util import synth_volume

from aind_exaspim_image_compression.utils import dct_quant as Q

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(8, 8, 8), (16, 24, 40), (20, 17, 33), (9, 70, 15), (64, 64, 64),
                                   (1, 1, 1), (7, 8, 9)])
@pytest.mark.parametrize("q", [1.0, 8.0, 37.5])
def test_indices_and_reconstruction_bit_exact(oracle, shape, q):
    vol = synth_volume(shape, seed=sum(shape), as_u16=True)[0]
    vol.reshape(-1)[:: max(1, vol.size // 7)] = 65535        # saturating values
    want = oracle.dctq_forward(vol, q)
    got = Q.quantise(vol, q)
    assert got.shape == want.shape and got.dtype == np.int32
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(Q.reconstruct(got, shape, q), oracle.dctq_inverse(want, shape, q))


def test_large_indices_and_errors(oracle):
    vol = np.full((8, 8, 16), 65535, dtype=np.uint16)
    got = Q.quantise(vol, 0.5)                                  # DC = 65535 * 512^0.5 / 0.5: far beyond int16
    np.testing.assert_array_equal(got, oracle.dctq_forward(vol, 0.5))
    assert abs(int(got.max()) - round(65535 * 512 ** 0.5 / 0.5)) <= 2 and np.count_nonzero(got) == 2
    # such indices are escapes of the rate model: 32 raw bits each on top of the symbol entropy
    assert Q.entropy_bits_per_voxel(got, vol.size) == pytest.approx(2 * 32.0 / vol.size + (
        -(2 / got.size) * np.log2(2 / got.size) - (1 - 2 / got.size) * np.log2(1 - 2 / got.size))
        * got.size / vol.size)
    with pytest.raises(ValueError):
        Q.quantise(vol, 0.0)
    with pytest.raises(ValueError):
        Q.quantise(vol[0], 1.0)
    with pytest.raises(ValueError):
        Q.reconstruct(got, (8, 8, 8), 1.0)


def test_rate_distortion_properties_at_scale():
    """256 x 264 x 272 (no oracle): error bounded by the step, rate falls and error grows with the
    step, a step of 1 reproduces the volume to within one count."""
    vol = synth_volume((64, 64, 64), seed=5, as_u16=True)[0]
    vol = np.tile(vol, (4, 5, 5))[:256, :264, :272].copy()
    last_bits, last_mae = None, None
    for q in (1.0, 4.0, 16.0, 64.0):
        rd = Q.rate_distortion(vol, q)
        # |rec - vol| <= sum of 512 coefficient errors of q/2 spread by an orthonormal basis + rounding
        assert rd["lmax"] <= 0.5 * q * np.sqrt(512.0) + 1.0
        if last_bits is not None:
            assert rd["bits_per_voxel"] < last_bits and rd["mae"] > last_mae
        last_bits, last_mae = rd["bits_per_voxel"], rd["mae"]
    idx = Q.quantise(vol, 1.0)
    rec = Q.reconstruct(idx, vol.shape, 1.0)
    assert np.abs(rec.astype(np.int32) - vol.astype(np.int32)).max() <= 1
    rd1 = Q.rate_distortion(vol, 1.0)
    assert Q.entropy_bits_per_voxel(idx, vol.size) == pytest.approx(rd1["order0_bits_per_voxel"])
    # the rate is the size of real byte streams, and the context coder beats the memoryless bound
    from aind_exaspim_image_compression.utils.chunk_codec import ExacCodec
    enc = ExacCodec(4).encode_volume(idx.reshape(-1, 8, 64), chunk=Q.INDEX_CHUNK)
    assert enc.nbytes == rd1["coded_bytes"] and rd1["bits_per_voxel"] < rd1["order0_bits_per_voxel"]
