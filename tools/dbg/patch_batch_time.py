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
    d_in, d_out = ctx.to_device(raw), ctx.alloc(raw.nbytes)
    ev = [ctx.event(), ctx.event()]
    ctx.set_option("profile", 1)
    for carry in (0, 1, 0, 1):
        ctx.set_option("bm_carry", carry)
        t0 = time.perf_counter(); out = denoise_patches(raw, 24.0); dt = time.perf_counter() - t0
        ctx.record(ev[0]); ctx.denoise_f32(d_in, d_out, (64, 64, 64), 24.0, batch=n, clip=(0.0, 65535.0)); ctx.record(ev[1]); ctx.sync()
        ph = ctx.profile_read()
        print(f"{n:5d} x 64^3, bm_carry {carry}: {dt * 1e3:8.1f} ms host to host; on the device {ctx.elapsed_ms(ev[0], ev[1]):7.1f} ms "
              f"= {n * 64 ** 3 / ctx.elapsed_ms(ev[0], ev[1]) / 1e3:7.1f} Mvoxels/s (block matching {ph.get('blockmatch_ht', 0):.1f} + {ph.get('blockmatch_wie', 0):.1f} ms)", flush=True)
    ctx.set_option("bm_carry", 1)
    ctx.set_option("profile", 0)
    d_in.free(); d_out.free()
