"""Block matching alone (no stages -- the probe flags give wrong tables) under several bm_march values:
    python tools/dbg/bm_march_probe.py edge march [march ...]
march = n blocks per segment (0: one tile per workgroup, 1: automatic), + 64: wave 0 skips the carry read,
+ 128: wave 7 skips the carry write (timing probes)."""
import os, sys
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import _native
import bench
n = int(sys.argv[1])
ctx = _native.context(0)
vol = bench.synth_u16((n,) * 3, 1000)
d_u16 = ctx.to_device(vol)
d_f32 = ctx.to_device(vol.astype(np.float32) - np.float32(37.0))
g = len(_native.grid_positions(n))
keys = ctx.alloc(g ** 3 * 16 * 4)
p = _native.default_params()
ev = [ctx.event(), ctx.event()]
for m in sys.argv[2:]:
    ctx.set_option("bm_march", int(m))
    out = []
    for name, fn in (("u16 (incl. conversion)", lambda: ctx.blockmatch_u16(d_u16, (n,) * 3, 24.0, p.c_match_ht, keys, p)),
                     ("f32", lambda: ctx.blockmatch(d_f32, (n,) * 3, 24.0, p.c_match_wie, keys, p))):
        best = 1e9
        for it in range(2):
            ctx.record(ev[0]); fn(); ctx.record(ev[1]); ctx.sync()
            best = min(best, ctx.elapsed_ms(ev[0], ev[1]))
        out.append(f"{name} {best:.1f} ms")
    print(f"bm_march {m:>4}: " + ", ".join(out), flush=True)
