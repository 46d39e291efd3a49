import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd")); sys.path.insert(0, os.path.join(R, "tests"))
from aind_exaspim_image_compression import _native as nat
from oracle import bm4d_oracle as O
from util import synth_volume
np.set_printoptions(precision=3, linewidth=200, suppress=True)
ctx = nat.context(0)
shape=(8,8,8)
noisy,_ = synth_volume(shape, seed=11)
keys = O.blockmatch(noisy, 24.0, 3.0)
num_w, den_w = O.stage(noisy, keys, 24.0)
d_n = ctx.to_device(noisy); d_k = ctx.to_device(keys)
d_num = ctx.alloc(noisy.nbytes).zero(); d_den = ctx.alloc(noisy.nbytes).zero()
ctx.stage(d_n, None, d_k, shape, 24.0, d_num, d_den); ctx.sync()
den = d_den.download(shape, np.float32)
r = den/den_w
print("keys", keys.ravel()[:4])
print("ratio by z (mean over y,x):", r.mean(axis=(1,2)))
print("ratio by y:", r.mean(axis=(0,2)))
print("ratio by x:", r.mean(axis=(0,1)))
print("ratio[0]:\n", r[0])
