"""Writes tests/golden/exac_v1.npz: inputs and EXAC v1 byte strings that pin the chunk coder's
format across rounds.  The format is this repo's specification (the reference's
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


if __name__ == "__main__":
    out = {}
    for name, arr in cases():
        b = co.encode(arr)
        back, used = co.decode(b, arr.size, arr.dtype.itemsize)
        assert used == len(b) and np.array_equal(back, arr.reshape(-1))
        out[name + "_in"] = arr
        out[name + "_bytes"] = np.frombuffer(b, dtype=np.uint8)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exac_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
