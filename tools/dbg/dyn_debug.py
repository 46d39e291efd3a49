import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd")); sys.path.insert(0, os.path.join(R, "tests"))
from aind_exaspim_image_compression import _native as nat
from oracle import bm4d_oracle as O
from util import synth_volume
ctx = nat.context(0)
shape=(40,44,48)
noisy,_ = synth_volume(shape, seed=11)
keys = O.blockmatch(noisy, 24.0, 3.0)
num_w, den_w = O.stage(noisy, keys, 24.0)
for trial in range(3):
    d_n = ctx.to_device(noisy); d_k = ctx.to_device(keys)
    d_num = ctx.alloc(noisy.nbytes).zero(); d_den = ctx.alloc(noisy.nbytes).zero()
    ctx.stage(d_n, None, d_k, shape, 24.0, d_num, d_den); ctx.sync()
    den = d_den.download(shape, np.float32)
    ratio = den/den_w
    bad = np.abs(ratio-1) > 1e-4
    print("trial", trial, "bad frac", bad.mean(), "sum ratio", den.sum()/den_w.sum())
    print(" bad per z-plane:", [int(bad[z].sum()) for z in range(shape[0])])
    zz,yy,xx = np.nonzero(bad)
    if len(zz): print(" y range", yy.min(), yy.max(), "x range", xx.min(), xx.max(), "ratio examples", ratio[bad][:6])
