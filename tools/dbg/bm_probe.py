"""Block matching alone (no stages) at edge^3, integer and fp32 kernel:
    python tools/dbg/bm_probe.py [edge=1024] [option=value ...]     (exabm4d_set_option pairs)"""
import os, sys
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import _native
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = _native.context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
vol = bench.synth_u16((n,) * 3, 1000)
d_u16 = ctx.to_device(vol)
d_f32 = ctx.to_device(vol.astype(np.float32) - np.float32(37.0))
g = len(_native.grid_positions(n))
keys = ctx.alloc(g ** 3 * 16 * 4)
p = _native.default_params()
ev = [ctx.event(), ctx.event()]
out = []
sums = []
for name, fn in (("u16 (incl. conversion)", lambda: ctx.blockmatch_u16(d_u16, (n,) * 3, 24.0, p.c_match_ht, keys, p)),
                 ("f32", lambda: ctx.blockmatch(d_f32, (n,) * 3, 24.0, p.c_match_wie, keys, p))):
    best = 1e9
    for it in range(3):
        ctx.record(ev[0]); fn(); ctx.record(ev[1]); ctx.sync()
        best = min(best, ctx.elapsed_ms(ev[0], ev[1]))
    out.append(f"{name} {best:.1f} ms")
    k = keys.download((g ** 3 * 16,), np.uint32)
    sums.append(int(k[::977].astype(np.uint64).sum()))       # a cheap fingerprint of the tables
print(" ".join(sys.argv[2:]) or "default", "|", ", ".join(out), "| table fingerprints", sums, flush=True)
