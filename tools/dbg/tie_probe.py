"""Where do the HIP pipeline's uint16 results differ from the oracle's on the volume that tripped
tools/fuzz_parity.py in round 2 (tests/golden/fuzz_tie_volume.npz)?  Prints, per (sigma, offset),
the differing fraction and how far the oracle's pre-rounding fp32 value sits from a half-integer
at the differing voxels, in counts and in ulps of the largest voxel of the block neighbourhood."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "aind-exaspim-image-compression_amd")]
from aind_exaspim_image_compression import _native  # noqa: E402
from oracle import bm4d_oracle as O  # noqa: E402

vol = np.load(os.path.join(ROOT, "tests", "golden", "fuzz_tie_volume.npz"))["vol"]
ctx = _native.context(0)
d_in, d_out = ctx.to_device(vol), ctx.alloc(vol.nbytes)
for sigma in (8.0, 24.0, 110.0):
    for offset in (0.0, 37.0, 100.5, 36.73):
        ctx.denoise_u16(d_in, d_out, vol.shape, sigma, offset)
        got = d_out.download(vol.shape, np.uint16).astype(np.int64)
        f = vol.astype(np.float32) - np.float32(offset)
        pre = O.bm4d(f, sigma).astype(np.float32) + np.float32(offset)
        want = np.rint(np.clip(pre, 0, 65535)).astype(np.int64)
        assert np.array_equal(want, O.bm4d_u16(vol, sigma, offset).astype(np.int64))
        diff = got != want
        tie = np.abs(pre - np.floor(pre) - np.float32(0.5))
        td = tie[diff]
        print(f"sigma {sigma:5.1f} offset {offset:6.2f}: differing {diff.mean():.2e} max|d| "
              f"{np.abs(got - want).max()}  tie distance at differing voxels: max {td.max() if td.size else 0:.3e} "
              f"median {np.median(td) if td.size else 0:.3e}; |pre| at differing: median "
              f"{np.median(np.abs(pre[diff])) if td.size else 0:.0f}; exact ties overall {np.mean(tie == 0):.2e}; "
              f"exact-tie share of differing {np.mean(td == 0) if td.size else 0:.2f}", flush=True)
