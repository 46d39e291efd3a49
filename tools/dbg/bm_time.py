"""time exabm4d_blockmatch_u16_dev alone (no stages): python tools/dbg/bm_time.py [edge]"""
import os, sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import _native
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = _native.context(0)
vol = bench.synth_u16((n,) * 3, 1000)
d_in = ctx.to_device(vol)
g = len(_native.grid_positions(n))
keys = ctx.alloc(g ** 3 * 16 * 4)
p = _native.default_params()
ev = [ctx.event(), ctx.event()]
for it in range(3):
    ctx.record(ev[0])
    ctx.blockmatch_u16(d_in, (n,) * 3, 24.0, p.c_match_ht, keys, p)
    ctx.record(ev[1]); ctx.sync()
    print(os.environ.get("EXABM4D_LIB", "default")[-24:], "blockmatch_u16 incl. conversion: %.1f ms" % ctx.elapsed_ms(ev[0], ev[1]), flush=True)
