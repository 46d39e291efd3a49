"""pairs vs one-wave stage kernels on the edge-heavy stress volumes (one stage), for bisecting"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import _native
import bench
ctx = _native.context(0)
vols = {"odd": bench.synth_u16((253, 251, 249), 7)}
v = bench.synth_u16((256,) * 3, 5); v[:64] = 0; v[:, :, -51:] = 0; vols["zeropad"] = v
for name, vol in vols.items():
    outs = []
    for pairs in (1, 0):
        ctx.set_option("stage_pairs", pairs)
        d_in = ctx.to_device(vol); d_out = ctx.alloc(vol.nbytes)
        ctx.denoise_u16(d_in, d_out, vol.shape, 24.0, 37.0, stages=1); ctx.sync()
        outs.append(d_out.download(vol.shape, np.uint16)); d_in.free(); d_out.free()
    d = np.abs(outs[0].astype(np.int32) - outs[1].astype(np.int32))
    print(os.environ.get("EXABM4D_LIB", "default")[-22:], name, "max", d.max(), "n>1", int((d > 1).sum()))
