"""Writes tests/golden/exac_v1.npz and exac_v2.npz: inputs and EXAC byte strings that pin the chunk
coder's two formats across rounds (v1's file must come out byte-identical to
the one committed in round 2).  The format is this repo's specification (the reference's
codec is third-party Blosc-zstd, absent here -- parity with its byte counts is unpinned), so the
vectors come from the oracle restatement, oracle/exac_codec.c; run from the repo root:

    python tests/golden/make_exac_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import codec_oracle as co  # noqa: E402


def cases():
    rng = np.random.default_rng(20261004)
    yield "u16_smooth", np.clip(rng.normal(37, 2.5, (12, 20, 64)), 0, 65535).round().astype(np.uint16)
    a = np.clip(rng.normal(300, 120, 777), 0, 65535).round().astype(np.uint16)
    a[::50] = 65535
    yield "u16_ragged_wide", a
    yield "u16_const", np.full(130, 37, dtype=np.uint16)
    yield "u16_single", np.array([513], dtype=np.uint16)
    yield "u16_all_bytes", (np.arange(4096, dtype=np.uint32) * 16 + 5).astype(np.uint16)
    i = rng.normal(0, 1.5, (9, 512)).round().astype(np.int32)
    i[:, 0] = rng.integers(-70000, 70000, 9)
    yield "i32_indices", i
    yield "i32_extremes", np.array([0, -1, 1, -2 ** 30, 2 ** 30, 255, -256, 65536], dtype=np.int32)


def cases_v2():
    """v2 models the chunk as a 3-D array: shapes matter."""
    yield from cases()
    rng = np.random.default_rng(20261005)
    zz, yy, xx = np.meshgrid(np.arange(10), np.arange(24), np.arange(64), indexing="ij")
    smooth = 37 + 900 * np.exp(-((yy - 11.5) ** 2 + (xx - 30) ** 2) / 40.0) + 3 * zz
    yield "u16_structure_3d", np.clip(smooth + rng.normal(0, 2, smooth.shape), 0, 65535).round().astype(np.uint16)
    yield "u16_narrow_rows", np.clip(rng.normal(500, 30, (7, 9, 12)), 0, 65535).round().astype(np.uint16)
    yield "u16_tiny_planes", np.clip(rng.normal(90, 4, (40, 3, 5)), 0, 65535).round().astype(np.uint16)
    yield "u16_full_range", rng.integers(0, 65536, (3, 8, 64)).astype(np.uint16)
    yield "u16_zero", np.zeros((2, 4, 64), dtype=np.uint16)
    blk = rng.laplace(0, 6, (20, 8, 64)).round().astype(np.int32)
    blk[:, 0, 0] += rng.integers(-40000, 40000, 20)
    blk[3, 2, 5], blk[4, 1, 9] = -2 ** 31, 2 ** 31 - 1
    yield "i32_blocks", blk


if __name__ == "__main__":
    for version, gen in ((1, cases), (2, cases_v2)):
        out = {}
        for name, arr in gen():
            b = co.encode(arr, version=version)
            back, used = co.decode(b, arr.size, arr.dtype.itemsize)
            assert used == len(b) and np.array_equal(back, arr.reshape(-1))
            out[name + "_in"] = arr
            out[name + "_bytes"] = np.frombuffer(b, dtype=np.uint8)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exac_v%d.npz" % version)
        if version == 1 and os.path.exists(path):
            old = np.load(path)
            assert sorted(old.files) == sorted(out) and all(np.array_equal(old[k], out[k]) for k in out), \
                "EXAC v1 vectors changed"
            print("v1 vectors unchanged")
            continue
        np.savez_compressed(path, **out)
        print("wrote", path, {k: v.shape for k, v in out.items()})
