"""Randomised differential run, HIP path against the oracle (a tool, not a test: run it on a GPU box
under `timeout`):

    python tools/fuzz_parity.py [seconds=120] [seed=0] [max_edge=72]

Every iteration draws a shape (8..72 per axis, ragged and even / odd row lengths), a data regime
(structure + noise, white noise, extremes 0 / 65535, constant, sparse), sigma and offset, and checks
  * stage-1 match tables of the uint16 entry point: bit-exact against the oracle,
  * the two-stage uint16 pipeline: within one count of the oracle on (almost) every voxel,
  * the chunk coder on a random chunk grid: bytes identical to the C restatement, exact decode.
Prints one line per iteration and a summary; exit code 1 on the first mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from aind_exaspim_image_compression import _native  # noqa: E402
from aind_exaspim_image_compression.utils.chunk_codec import ExacCodec  # noqa: E402
from oracle import bm4d_oracle as O  # noqa: E402
from oracle import codec_oracle as C  # noqa: E402
from util import synth_volume  # noqa: E402


def draw_volume(rng, shape):
    kind = rng.integers(0, 6)
    if kind == 0:
        v = synth_volume(shape, seed=int(rng.integers(1 << 30)), as_u16=True)[0]
        name = "structure"
    elif kind == 1:
        v = rng.integers(0, 65536, shape).astype(np.uint16)
        name = "white"
    elif kind == 2:
        v = synth_volume(shape, seed=int(rng.integers(1 << 30)), as_u16=True)[0]
        v.reshape(-1)[:: int(rng.integers(7, 40))] = 65535
        v.reshape(-1)[3:: int(rng.integers(7, 40))] = 0
        name = "extremes"
    elif kind == 3:
        v = np.full(shape, int(rng.integers(0, 65536)), np.uint16)
        name = "constant"
    elif kind == 4:
        v = np.clip(rng.normal(40, 24, shape), 0, 65535).astype(np.uint16)
        z, y, x = (int(rng.integers(0, s)) for s in shape)
        v[z: z + 9, y: y + 11, x: x + 7] = int(rng.integers(2000, 60000))
        name = "sparse"
    else:
        v = np.clip(rng.normal(30000, 3000, shape), 0, 65535).astype(np.uint16)
        name = "bright"
    return name, v


def unexplained_voxels(ctx, vol, sigma, offset, got, ref):
    """The account tests/test_pipeline_differences_gpu.py gives of a difference, applied to one volume: the
    GPU's stage-2 match tables on ITS basic estimate against the oracle's on the oracle's; a voxel where the
    two uint16 results differ must lie inside the aggregation footprint of a reference block whose group
    changed, or on a rounding near-tie of the oracle's estimate.  Returns (voxels without such an account,
    share of groups that changed)."""
    shape, n = vol.shape, vol.size
    f = vol.astype(np.float32) - np.float32(offset)
    basic_o = O.bm4d(f, sigma, stages=1).astype(np.float32)
    pre_o = O.bm4d(f, sigma, stages=2).astype(np.float32) + np.float32(offset)
    d_in, d_out = ctx.to_device(f), ctx.alloc(4 * n)
    ctx.denoise_f32(d_in, d_out, shape, sigma, stages=1)
    ctx.sync()
    basic_g = d_out.download(shape, np.float32)
    g = [len(_native.grid_positions(m)) for m in shape]
    d_keys = ctx.alloc(g[0] * g[1] * g[2] * 64)
    keys = {}
    for name, basic in (("gpu", basic_g), ("oracle", basic_o)):
        d_in.upload(basic)
        ctx.blockmatch(d_in, shape, sigma, _native.default_params().c_match_wie, d_keys)
        ctx.sync()
        keys[name] = d_keys.download((g[0], g[1], g[2], 16), np.uint32)
    for b in (d_in, d_out, d_keys):
        b.free()
    # same basic estimate -> the GPU's stage-2 tables are the oracle's, bit for bit
    if not np.array_equal(keys["oracle"], O.blockmatch(basic_o, sigma, _native.default_params().c_match_wie)):
        return -1, 0.0
    changed = np.any((keys["gpu"] & 0x7FF) != (keys["oracle"] & 0x7FF), axis=-1)
    foot = np.zeros(shape, bool)
    pz, py, px = (_native.grid_positions(m) for m in shape)
    for iz, iy, ix in zip(*np.nonzero(changed)):
        z, y, x = int(pz[iz]), int(py[iy]), int(px[ix])
        foot[max(0, z - 5):z + 13, max(0, y - 5):y + 13, max(0, x - 5):x + 13] = True
    tie = np.abs(pre_o - np.floor(pre_o) - np.float32(0.5)) <= 2e-5 * 65535.0
    diff = got.astype(np.int64) != ref.astype(np.int64)
    return int((diff & ~foot & ~tie).sum()), float(changed.mean())


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    max_edge = int(sys.argv[3]) if len(sys.argv) > 3 else 72
    ctx = _native.context(0)
    codec = ExacCodec()
    t0, it = time.time(), 0
    while time.time() - t0 < budget:
        it += 1
        shape = tuple(int(rng.integers(8, max_edge + 1)) for _ in range(3))
        if rng.random() < 0.3:
            shape = shape[:2] + (shape[2] // 2 * 2,)
        name, vol = draw_volume(rng, shape)
        sigma = float(rng.choice([8.0, 16.0, 24.0, 60.0, 110.0]))
        offset = float(rng.choice([0.0, 37.0, 100.5]))
        f = vol.astype(np.float32) - np.float32(offset)
        # block matching's launch variants: carry between tiles forced / off, workgroup order (tables must not care)
        ctx.set_option("bm_carry", int(rng.choice([0, 2, 2])))
        ctx.set_option("bm_xcd_mode", int(rng.integers(0, 4)))

        # stage-1 tables (uint16 entry point: integer kernel where it applies)
        g = [len(_native.grid_positions(n)) for n in shape]
        d_vol = ctx.to_device(vol)
        d_keys = ctx.alloc(g[0] * g[1] * g[2] * 64)
        ctx.blockmatch_u16(d_vol, shape, sigma, 3.0, d_keys)
        ctx.sync()
        keys = d_keys.download((g[0], g[1], g[2], 16), np.uint32)
        want = O.blockmatch(vol.astype(np.float32), sigma, 3.0)
        ok_keys = np.array_equal(keys, want)

        # two-stage uint16 pipeline
        d_out = ctx.alloc(vol.nbytes)
        ctx.denoise_u16(d_vol, d_out, shape, sigma, offset)
        got = d_out.download(shape, np.uint16)
        ref = O.bm4d_u16(vol, sigma, offset)
        d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
        # one count where the two fp32 summation orders round apart, or where the last bits of the
        # basic estimate moved a stage-2 match table (tests/test_pipeline_differences_gpu.py pins
        # both mechanisms; volumes with isolated 0 / 65535 voxels reach 0.5 % by the second one,
        # whatever the offset -- tools/dbg/tie_probe.py).  A changed group can move a voxel by more
        # than a count (seen: 2 at sigma 110 on an "extremes" volume, 3 in bench.py's psnr block);
        # the aggregation order varies between launches, so the same input does not always show it
        ok_pipe = d.max() <= 3 and np.mean(d > 1) < 1e-4 and np.mean(d > 0) < 2e-2
        note = ""
        if not ok_pipe and np.mean(d > 0) < 2e-2:
            # beyond the usual bound (seen: 4 counts next to a 60000-count box in noise): acceptable only
            # if every differing voxel is accounted for by a changed stage-2 group or a rounding near-tie
            left, share = unexplained_voxels(ctx, vol, sigma, offset, got, ref)
            ok_pipe = left == 0
            note = f" [groups changed {share:.2%}, voxels without an account {left}]"
        del f

        # chunk coder on the denoised volume, random chunk grid
        chunk = tuple(int(rng.choice([8, 16, 24, 64])) for _ in range(3))
        enc = codec.encode_volume(got, chunk=chunk)
        ok_codec = True
        k = 0
        for piece in C.chunks(got, chunk):
            if enc.chunk_bytes(k) != bytes(C.encode(piece)):
                ok_codec = False
                break
            k += 1
        ok_codec = ok_codec and np.array_equal(codec.decode_volume(enc), got)
        for b in (d_vol, d_keys, d_out):
            b.free()
        print(f"{it:4d} {name:9s} {str(shape):14s} sigma {sigma:5.1f} offset {offset:5.1f} chunk {chunk} "
              f"keys {'ok' if ok_keys else 'MISMATCH'} pipeline max|d| {int(d.max())} "
              f"frac {float(np.mean(d > 0)):.1e} beyond one {float(np.mean(d > 1)):.1e} {'ok' if ok_pipe else 'MISMATCH'}{note} "
              f"codec {'ok' if ok_codec else 'MISMATCH'}", flush=True)
        if not (ok_keys and ok_pipe and ok_codec):
            np.save(os.path.join(ROOT, "gpurun_out", "fuzz_fail_vol.npy"), vol)
            print("FAILED", flush=True)
            return 1
    print(f"{it} iterations, all agree", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
