"""Latency of small batched calls (the broker's regime): b patches of 64^3 fp32, device-resident, per phase.
python tools/dbg/small_batch_time.py [b ...]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "aind-exaspim-image-compression_amd"), os.path.join(R, "tests")]
from aind_exaspim_image_compression import _native
from util import synth_volume
ctx = _native.context(0)
base = np.stack([synth_volume((64,) * 3, seed=i)[0] for i in range(8)])
ctx.set_option("profile", 1)
if os.environ.get("EXABM4D_ZERO_OVERLAP"):
    ctx.set_option("zero_overlap", int(os.environ["EXABM4D_ZERO_OVERLAP"]))
if os.environ.get("EXABM4D_PROFILE"):
    ctx.set_option("profile", int(os.environ["EXABM4D_PROFILE"]))
for b in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 256]:
    raw = np.concatenate([base] * ((b + 7) // 8))[:b].copy()
    d_in, d_out = ctx.to_device(raw), ctx.alloc(raw.nbytes)
    ev = [ctx.event(), ctx.event()]
    best, ph_best = 1e9, None
    for rep in range(5):
        ctx.record(ev[0]); ctx.denoise_f32(d_in, d_out, (64, 64, 64), 24.0, batch=b, clip=(0.0, 65535.0)); ctx.record(ev[1]); ctx.sync()
        ms = ctx.elapsed_ms(ev[0], ev[1])
        if ms < best:
            best, ph_best = ms, (ctx.profile_read() if not os.environ.get("EXABM4D_PROFILE") else {})
    t0 = time.perf_counter()
    for rep in range(5):
        ctx.denoise_f32(d_in, d_out, (64, 64, 64), 24.0, batch=b, clip=(0.0, 65535.0)); ctx.sync()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    print(f"b {b:4d}: device {best:7.3f} ms = {best / b:6.3f} per patch; wall incl. launch + sync {wall:7.3f} ms; "
          + " ".join(f"{k.replace('blockmatch', 'bm').replace('normalize', 'nrm')} {v:.2f}" for k, v in ph_best.items() if v > 0.005), flush=True)
    d_in.free(); d_out.free()
