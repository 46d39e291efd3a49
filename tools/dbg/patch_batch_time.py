"""denoise_patches (the precompute.py work shape: N patches of 64^3, fp32 in / out, host to host) with and
without block matching's carry between tiles:  python tools/dbg/patch_batch_time.py [n ...]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "aind-exaspim-image-compression_amd"), os.path.join(R, "tests")]
from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.bm4d import denoise_patches
from util import synth_volume
ctx = _native.context(0)
base = np.stack([synth_volume((64,) * 3, seed=i)[0] for i in range(8)])
for n in [int(v) for v in sys.argv[1:]] or [200, 1000]:
    raw = np.concatenate([base] * ((n + 7) // 8))[:n]
    for carry in (0, 1, 0, 1):
        ctx.set_option("bm_carry", carry)
        t0 = time.perf_counter(); out = denoise_patches(raw, 24.0); dt = time.perf_counter() - t0
        print(f"{n:5d} x 64^3, bm_carry {carry}: {dt * 1e3:8.1f} ms host to host = {n * 64 ** 3 / dt / 1e6:7.1f} Mvoxels/s", flush=True)
    ctx.set_option("bm_carry", 1)
