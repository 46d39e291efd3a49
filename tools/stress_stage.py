"""Stress the stage kernels on volumes with every group-size regime at multi-layer scale: the CPU
port's uint16 volume (bit-identical to the oracle's, tests/test_oracle_bm4d.py), voxel for voxel, from two
launch shapes (automatic z chunks; one chunk per tile column).  Round 4: exact equality (integer
aggregation sums); rounds 2-3 compared with the one-wave-per-group kernels within a count.

usage: python tools/stress_stage.py [edge=192]   (wrap in `timeout`: a hang is the failure mode; the CPU
port needs about a second per 10^7 voxels and stage on 16 threads)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
sys.path.insert(0, ROOT)

from aind_exaspim_image_compression import _native  # noqa: E402
import bench  # noqa: E402


def volumes(edge):
    rng = np.random.default_rng(0)
    for seed in (1, 2, 3):
        yield f"synthetic seed {seed}", bench.synth_u16((edge,) * 3, seed)
    v = bench.synth_u16((edge,) * 3, 4).astype(np.int64)
    v[:, edge // 3: edge // 2, :] += rng.integers(0, 6000, (edge, edge // 2 - edge // 3, edge))
    yield "white-noise band (one-block groups)", np.clip(v, 0, 65535).astype(np.uint16)
    v = bench.synth_u16((edge,) * 3, 5)
    v[: edge // 4] = 0
    v[:, :, -edge // 5:] = 0
    yield "zero padding", v
    v = np.zeros((edge,) * 3, np.uint16)
    v[edge // 2 - 20: edge // 2 + 20, 10:50, 30:90] = 5000
    yield "mostly empty", v
    v = rng.integers(0, 65536, (edge,) * 3).astype(np.uint16)
    yield "full-range white noise", v
    yield "thin slab", bench.synth_u16((24, edge, edge), 6)
    yield "odd extents", bench.synth_u16((edge - 3, edge - 5, edge - 7), 7)


def main():
    from oracle import bm4d_oracle as O
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    ctx = _native.context(0)
    for name, vol in volumes(edge):
        line = f"{name:38s} {str(vol.shape):16s}"
        for stages in (1, 2):
            want = O.bm4d_u16(vol, bench.SIGMA, bench.OFFSET, stages=stages, port=True)
            for chunks in (0, 1):
                ctx.set_option("stage_chunks", chunks)
                d_in = ctx.to_device(vol)
                d_out = ctx.alloc(vol.nbytes)
                t0 = time.perf_counter()
                ctx.denoise_u16(d_in, d_out, vol.shape, bench.SIGMA, bench.OFFSET, stages=stages)
                ctx.sync()
                dt = time.perf_counter() - t0
                got = d_out.download(vol.shape, np.uint16)
                d_in.free()
                d_out.free()
                ok = np.array_equal(got, want)
                line += f"  | {stages} stage{'s' if stages > 1 else ' '} chunks {chunks}: {dt*1e3:6.1f} ms {'OK' if ok else 'MISMATCH'}"
                if not ok:
                    print(line, flush=True)
                    bad = np.argwhere(got != want)
                    print("   first mismatches (z, y, x):", bad[:6].tolist(), "z range", bad[:, 0].min(), bad[:, 0].max(),
                          "count", len(bad), flush=True)
                    if os.environ.get("STRESS_KEEP_GOING") != "1":
                        sys.exit(1)
        print(line, flush=True)
    ctx.set_option("stage_chunks", 0)
    print("all volumes agree")


if __name__ == "__main__":
    main()
