import os, sys, ctypes, numpy as np
os.environ["EXABM4D_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libexabm4d_bmstamps.so")
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import _native as nat
import bench
ctx = nat.context(0)
shape = (256,)*3
vol = bench.synth_u16(shape, 1000).astype(np.float32)
d = ctx.to_device(vol); g=[len(nat.grid_positions(n)) for n in shape]
k = ctx.alloc(g[0]*g[1]*g[2]*64)
L = ctypes.CDLL(os.environ["EXABM4D_LIB"])
out = (ctypes.c_ulonglong*8)()
for i in range(2):
    ctx.blockmatch(d, shape, 24.0, 3.0, k); ctx.sync()
    L.exabm4d_debug_bm_stamps(out)
names=["wait_dma","wait_A","issue_dma","compute","exchange","total","first_barrier"]
print({n: round(out[i]/out[5],3) for i,n in enumerate(names)}, out[5])
