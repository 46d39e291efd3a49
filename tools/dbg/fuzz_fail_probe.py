"""Replay gpurun_out/fuzz_fail_vol.npy (saved by tools/fuzz_parity.py) with the account of differences:
    python tools/dbg/fuzz_fail_probe.py sigma offset [volume.npz]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "aind-exaspim-image-compression_amd"), os.path.join(R, "tests"), os.path.join(R, "tools")]
from aind_exaspim_image_compression import _native
import fuzz_parity as F
vol = np.load(sys.argv[3])["vol"] if len(sys.argv) > 3 else np.load(os.path.join(R, "gpurun_out", "fuzz_fail_vol.npy"))
sigma, offset = float(sys.argv[1]), float(sys.argv[2])
ctx = _native.context(0)
ref = F.O.bm4d_u16(vol, sigma, offset)
for carry in (0, 2):
    for rep in range(3):
        ctx.set_option("bm_carry", carry)
        d_vol, d_out = ctx.to_device(vol), ctx.alloc(vol.nbytes)
        ctx.denoise_u16(d_vol, d_out, vol.shape, sigma, offset)
        got = d_out.download(vol.shape, np.uint16)
        d_vol.free(); d_out.free()
        d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
        left, share = F.unexplained_voxels(ctx, vol, sigma, offset, got, ref)
        print(f"carry {carry} run {rep}: max|d| {int(d.max())} differing {np.mean(d > 0):.2e} beyond one {np.mean(d > 1):.2e}; "
              f"groups changed {share:.2%}; voxels without an account {left}", flush=True)
