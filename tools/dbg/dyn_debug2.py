import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd")); sys.path.insert(0, os.path.join(R, "tests"))
from aind_exaspim_image_compression import _native as nat
from oracle import bm4d_oracle as O
from util import synth_volume
ctx = nat.context(0)
for shape in [(8,8,8),(8,8,12),(8,12,12),(12,12,12),(8,8,24),(8,24,24),(8,20,8),(24,8,8)]:
    noisy,_ = synth_volume(shape, seed=11)
    keys = O.blockmatch(noisy, 24.0, 3.0)
    num_w, den_w = O.stage(noisy, keys, 24.0)
    d_n = ctx.to_device(noisy); d_k = ctx.to_device(keys)
    d_num = ctx.alloc(noisy.nbytes).zero(); d_den = ctx.alloc(noisy.nbytes).zero()
    ctx.stage(d_n, None, d_k, shape, 24.0, d_num, d_den); ctx.sync()
    den = d_den.download(shape, np.float32); num = d_num.download(shape, np.float32)
    r = den/den_w
    print(shape, "grid", keys.shape[:3], "den ratio min/max", r.min(), r.max(), "est maxdiff", np.abs(num/den - num_w/den_w).max())
