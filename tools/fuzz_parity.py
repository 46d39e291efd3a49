"""Randomised differential run, HIP path against the oracle (a tool, not a test: run it on a GPU box
under `timeout`):

    python tools/fuzz_parity.py [seconds=120] [seed=0] [max_edge=72]

Every iteration draws a shape (8..72 per axis, ragged and even / odd row lengths), a data regime
(structure + noise, white noise, extremes 0 / 65535, constant, sparse), sigma and offset, and checks
  * stage-1 match tables of the uint16 entry point: bit-exact against the oracle,
  * the two-stage uint16 pipeline: the oracle's uint16 volume, voxel for voxel (round 4: the aggregation
    sums are integers; rounds 1-3 allowed a count, and an "account" of changed stage-2 groups beyond it),
    also with the stage kernels' launch shape drawn at random (z chunks, tile order, gather form) and on
    a second launch,
  * the fp32 entry point on the same counts at a random scale (2^-40 ... 2^39, the top of the working
    range of DESIGN.md 3.8: the numerator's unit follows the data), one or two stages, clipped or not:
    the oracle's fp32 volume, bit for bit,
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
        offset = float(rng.choice([0.0, 37.0, 100.5, 36.73]))
        f = vol.astype(np.float32) - np.float32(offset)
        # block matching's launch variants: carry between tiles forced / off, workgroup order (tables must not care)
        ctx.set_option("bm_carry", int(rng.choice([0, 2, 2])))
        ctx.set_option("bm_xcd_mode", int(rng.integers(0, 4)))
        # ... and the stage kernels' (sums must not care)
        ctx.set_option("stage_chunks", int(rng.choice([0, 0, 1, 3])))
        ctx.set_option("stage_strip", int(rng.choice([0, 2, 3])))
        ctx.set_option("stage_pairvol", int(rng.integers(0, 2)))

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
        ctx.denoise_u16(d_vol, d_out, shape, sigma, offset)                 # again: a function of its input
        ok_pipe = not d.any() and np.array_equal(d_out.download(shape, np.uint16), got)
        note = ""
        # the fp32 entry point on the same counts at a random scale: every volume carries its own unit
        # (E from its largest |v|) -- one stage or two, clipped or not
        scale = np.float32(rng.choice([1.0, 1.0 / 4096.0, 3.0e4, 2.0 ** -40, 2.0 ** 39]))
        g32 = (f * scale).astype(np.float32)
        st = int(rng.integers(1, 3))
        clip = None if rng.random() < 0.5 else (0.0, float(np.float32(20000.0) * scale))
        got32 = ctx.denoise_f32_host(g32, float(np.float32(sigma) * scale), stages=st, clip=clip)
        ok_f32 = np.array_equal(got32, O.bm4d(g32, float(np.float32(sigma) * scale), stages=st, clip=clip))
        del f, g32

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
              f"codec {'ok' if ok_codec else 'MISMATCH'} f32 x{float(scale):.3g} {'ok' if ok_f32 else 'MISMATCH'}", flush=True)
        if not (ok_keys and ok_pipe and ok_codec and ok_f32):
            np.save(os.path.join(ROOT, "gpurun_out", "fuzz_fail_vol.npy"), vol)
            print("FAILED", flush=True)
            return 1
    print(f"{it} iterations, all agree", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
