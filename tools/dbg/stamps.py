import os, sys, ctypes, numpy as np
os.environ["EXABM4D_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libexabm4d_stamps.so")
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import _native as nat
import bench
ctx = nat.context(0)
shape = (int(os.environ.get("SIZE", "512")),)*3
vol = bench.synth_u16(shape, 1000)
d_in = ctx.to_device(vol); d_out = ctx.alloc(vol.nbytes)
L = ctypes.CDLL(os.environ["EXABM4D_LIB"])
out = (ctypes.c_ulonglong*16)()
ctx.denoise_u16(d_in, d_out, shape, 24.0, 37.0); ctx.sync()
L.exabm4d_debug_stamps(out, 1)
ctx.denoise_u16(d_in, d_out, shape, 24.0, 37.0); ctx.sync()
L.exabm4d_debug_stamps(out, 1)
names = ["fwd","shrink","inv","wait+lock","scatter","closer_flush","final_barrier","total"]
if os.environ.get("PAIRS", "1") == "1":
    hn = ["fwd", "shrink+exchange(all)", "exchange_wait", "lock_wait", "rmw", "ack+layer_wait", "closer_flush", "total"]
    for base, lab in ((0, "HT-pairs"), (8, "WIE-pairs")):
        tot = out[base + 7]
        print(lab, {n: round(out[base + i]/tot,3) for i,n in enumerate(hn)}, tot)
for base,lab in ((0,"HT"),(8,"WIE")):
    tot = out[base+7]
    print(lab, {n: round(out[base+i]/tot,3) for i,n in enumerate(names)}, "total wave-cycles(100MHz ticks?)", tot)
