"""BASELINE.json config 4 -- chunk-local mode (cores + halo, the read window clamped to the volume,
every padded chunk denoised in isolation) on the GPU: match tables of the padded chunks and
the uint16 result bit-exact against the oracle that processes the identical padded arrays, ragged
chunk grids, slab-style core ranges with real neighbour planes, batching under a small scratch
budget, and the accuracy note (PSNR against whole-volume processing)."""
import numpy as np
import pytest

from util import psnr, synth_volume

from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.bm4d import denoise_chunked, denoise_volume

pytestmark = pytest.mark.gpu
SIGMA, OFFSET = 24.0, 37.0


def close_u16(got, want):
    """(The name is history: since round 4 -- integer aggregation sums -- "close" means equal.)"""
    np.testing.assert_array_equal(got, want)


def test_eight_chunks_match_tables_and_result(ctx, oracle):
    """48^3 volume, 24^3 cores, 8-voxel halo: eight padded 32^3 chunks (every chunk has three
    faces on the volume's boundary, where the window is cut) as ONE batched call."""
    vol, _ = synth_volume((48, 48, 48), seed=3, as_u16=True)
    chunks = list(oracle.padded_chunks(vol, 24, 8))
    assert len(chunks) == 8 and all(p.shape == (32, 32, 32) for _, _, p in chunks)
    # (a) block matching on the identical padded arrays, batched: keys bit-exact per chunk
    batch = np.stack([p.astype(np.float32) - np.float32(OFFSET) for _, _, p in chunks])
    d_vol = ctx.to_device(batch)
    g = len(_native.grid_positions(32))
    d_keys = ctx.alloc(8 * g ** 3 * 16 * 4)
    ctx.blockmatch(d_vol, (32, 32, 32), SIGMA, 3.0, d_keys, batch=8)
    ctx.sync()
    keys = d_keys.download((8, g, g, g, 16), np.uint32)
    d_vol.free()
    d_keys.free()
    for b in range(8):
        np.testing.assert_array_equal(keys[b], oracle.blockmatch(batch[b], SIGMA, 3.0))
    # (b) the whole chunk-local call
    got = denoise_chunked(vol, SIGMA, OFFSET, chunk=24, halo=8)
    close_u16(got, oracle.bm4d_u16_chunked(vol, SIGMA, OFFSET, 24, 8))
    # a chunk that covers the volume with halo 0 is the whole-volume pipeline
    np.testing.assert_array_equal(denoise_chunked(vol, SIGMA, OFFSET, chunk=64, halo=0),
                                  denoise_volume(vol, SIGMA, OFFSET))


def test_ragged_grid_small_budget_and_core_range(ctx, oracle):
    """Ragged last chunks on two axes, interior / face chunks of different padded shapes (twelve
    shape classes), batches of a few chunks each (tiny scratch budget), and a slab-style call: only planes [8, 40) are cores, the planes around them
    are real data (what a rank holds after the input-halo exchange)."""
    vol, _ = synth_volume((40, 52, 44), seed=8, as_u16=True)
    want = oracle.bm4d_u16_chunked(vol, SIGMA, OFFSET, 16, 4)
    ctx.set_option("chunk_budget_mb", 8)
    try:
        close_u16(denoise_chunked(vol, SIGMA, OFFSET, chunk=16, halo=4), want)
    finally:
        ctx.set_option("chunk_budget_mb", 32768)
    d_in = ctx.to_device(vol)
    d_out = ctx.alloc(32 * 52 * 44 * 2)
    ctx.denoise_chunked_u16(d_in, d_out, vol.shape, SIGMA, OFFSET, chunk=16, halo=8, core=(8, 40))
    ctx.sync()
    got = d_out.download((32, 52, 44), np.uint16)
    d_in.free()
    d_out.free()
    close_u16(got, oracle.bm4d_u16_chunked(vol, SIGMA, OFFSET, 16, 8, core=(8, 40)))
    with pytest.raises(ValueError):
        denoise_chunked(vol, SIGMA, OFFSET, chunk=33, halo=0)     # 40 = 33 + 7: a 7-voxel chunk is too thin
    with pytest.raises(ValueError):
        ctx.denoise_chunked_u16(0, 0, vol.shape, SIGMA, OFFSET)


def test_accuracy_note_256(ctx):
    """256^3, cores of 128 with the 8-voxel halo of config 4 against whole-volume processing:
    the PSNR difference against the clean volume is the accuracy cost of the short halo (exact
    equivalence needs 24 voxels per stage).  Recorded in DESIGN.md."""
    base, clean = synth_volume((64, 64, 64), seed=12, as_u16=True)
    rng = np.random.default_rng(5)
    cl = np.tile(clean, (4, 4, 4))
    vol = np.rint(np.clip(cl + rng.normal(0, SIGMA, cl.shape), 0, 65535)).astype(np.uint16)
    whole = denoise_volume(vol, SIGMA, OFFSET)
    chunked = denoise_chunked(vol, SIGMA, OFFSET, chunk=128, halo=8)
    peak = float(cl.max() - cl.min())
    p_whole, p_chunk = psnr(whole, cl, peak), psnr(chunked, cl, peak)
    print(f"PSNR vs clean: whole {p_whole:.3f} dB, 128^3+8 chunks {p_chunk:.3f} dB, "
          f"differing voxels {np.mean(whole != chunked):.4f}")
    assert abs(p_whole - p_chunk) < 0.1
    # >= 48 voxels from the faces BETWEEN chunks a voxel never sees the cut (the volume's own
    # faces are the same in both): identical
    inner = (slice(0, 80),) * 3
    np.testing.assert_array_equal(whole[inner], chunked[inner])


def test_config4_geometry_256_cores_plus_8(ctx):
    """BASELINE config 4's real chunk geometry -- 256^3 cores + 8-voxel halo -- as 2 x 2 x 2 cores of a
    512^3 volume (padded chunks of 264^3: one face per axis on the volume's boundary, one cut).
    Property (no oracle at this size): a voxel further than 48 from the cuts between cores never
    sees a cut in either stage (24 voxels of context per stage), so the chunk-local result equals
    whole-volume processing there exactly (integer sums, the same blocks); near the cuts it may differ, but
    stays a denoised volume (PSNR against the clean volume within 0.1 dB)."""
    base, clean = synth_volume((64, 64, 64), seed=21, as_u16=True)
    cl = np.tile(clean, (8, 8, 8)).astype(np.float32)
    rng = np.random.default_rng(9)
    vol = np.empty(cl.shape, np.uint16)
    for z in range(0, 512, 64):                      # noise in slabs: bounded host memory
        vol[z:z + 64] = np.rint(np.clip(cl[z:z + 64] + rng.standard_normal((64, 512, 512), dtype=np.float32)
                                        * np.float32(SIGMA), 0, 65535)).astype(np.uint16)
    whole = denoise_volume(vol, SIGMA, OFFSET)
    chunked = denoise_chunked(vol, SIGMA, OFFSET, chunk=256, halo=8)
    assert chunked.shape == vol.shape and chunked.dtype == np.uint16
    far = (slice(0, 256 - 48), slice(256 + 48, 512))
    for sz in far:
        for sy in far:
            for sx in far:
                np.testing.assert_array_equal(whole[sz, sy, sx], chunked[sz, sy, sx], err_msg=f"{sz} {sy} {sx}")
    near = np.abs(whole[248:264].astype(np.int32) - chunked[248:264].astype(np.int32))
    assert near.max() > 0                             # the cut is real: chunk-local semantics, not a no-op
    peak = float(cl.max() - cl.min())
    assert abs(psnr(whole, cl, peak) - psnr(chunked, cl, peak)) < 0.1


def test_streamed_host_volume_equals_the_one_call_result(ctx, oracle, tmp_path):
    """exabm4d_denoise_chunked_u16_host: the volume goes through the device one layer of chunks at a
    time (here 16 + 16 + 8 planes: three layers, the last one ragged, windows cut at both ends),
    copies of the neighbouring layers under the kernels.  Same chunks, same pipeline: the oracle's
    chunk-local result, and the one-call device result."""
    from aind_exaspim_image_compression.bm4d import denoise_chunked_streamed
    vol, _ = synth_volume((40, 36, 44), seed=21, as_u16=True)
    want = oracle.bm4d_u16_chunked(vol, SIGMA, OFFSET, 16, 4)
    got = denoise_chunked_streamed(vol, SIGMA, OFFSET, chunk=16, halo=4)
    assert got.dtype == np.uint16 and got.shape == vol.shape
    close_u16(got, want)
    close_u16(got, denoise_chunked(vol, SIGMA, OFFSET, chunk=16, halo=4))
    # file-backed source and destination (a raw tile on disk in, a raw tile on disk out)
    src = np.memmap(tmp_path / "in.u16", dtype=np.uint16, mode="w+", shape=vol.shape)
    src[:] = vol
    src.flush()
    src = np.memmap(tmp_path / "in.u16", dtype=np.uint16, mode="r", shape=vol.shape)
    dst = np.memmap(tmp_path / "out.u16", dtype=np.uint16, mode="w+", shape=vol.shape)
    assert denoise_chunked_streamed(src, SIGMA, OFFSET, chunk=16, halo=4, out=dst) is dst
    dst.flush()
    close_u16(np.fromfile(tmp_path / "out.u16", dtype=np.uint16).reshape(vol.shape), want)
    # one layer only (no second window), one stage, and a layer count that is odd / even
    one = denoise_chunked_streamed(vol, SIGMA, OFFSET, chunk=64, halo=0)
    np.testing.assert_array_equal(one, denoise_volume(vol, SIGMA, OFFSET))
    close_u16(denoise_chunked_streamed(vol, SIGMA, OFFSET, chunk=8, halo=8, stages=1),
              denoise_chunked(vol, SIGMA, OFFSET, chunk=8, halo=8, stages=1))


def test_streamed_host_volume_errors(ctx):
    from aind_exaspim_image_compression.bm4d import denoise_chunked_streamed
    vol, _ = synth_volume((24, 24, 24), seed=2, as_u16=True)
    with pytest.raises(ValueError):
        denoise_chunked_streamed(vol, SIGMA, OFFSET, out=vol)                       # in place
    with pytest.raises(ValueError):
        denoise_chunked_streamed(vol, SIGMA, OFFSET, out=np.empty((24, 24, 25), np.uint16))
    with pytest.raises((ValueError, RuntimeError)):
        denoise_chunked_streamed(vol, SIGMA, OFFSET, chunk=4, halo=1)               # padded chunk thinner than a block
    # the C entry point checks the overlap itself (a caller that is not the Python wrapper)
    import ctypes
    from aind_exaspim_image_compression import _native as nat
    p = nat.default_params()
    rc = nat.lib().exabm4d_denoise_chunked_u16_host(ctx.handle, vol.ctypes.data, vol.ctypes.data + 2 * 24 * 24, 20, 24, 24,
                                                     16, 8, SIGMA, OFFSET, ctypes.byref(p), 2)
    assert rc != 0 and b"overlap" in nat.lib().exabm4d_last_error(ctx.handle)
    # the context is usable afterwards
    close_u16(denoise_chunked_streamed(vol, SIGMA, OFFSET, chunk=16, halo=8),
              denoise_chunked(vol, SIGMA, OFFSET, chunk=16, halo=8))
