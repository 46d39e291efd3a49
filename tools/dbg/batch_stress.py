import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd")); sys.path.insert(0, os.path.join(R, "tests"))
from aind_exaspim_image_compression.bm4d import denoise_patches, bm4d
from util import synth_volume
rng = np.random.default_rng(0)
for n, edge in ((1, 64), (7, 64), (64, 64), (200, 64), (3, 54), (5, 96), (2, 160)):
    raw = np.stack([synth_volume((edge,) * 3, seed=i)[0] for i in range(min(n, 8))])
    raw = np.concatenate([raw] * ((n + len(raw) - 1) // len(raw)))[:n]
    t0 = time.time(); out = denoise_patches(raw, 24.0); dt = time.time() - t0
    assert out.shape == raw.shape and np.isfinite(out).all() and out.min() >= 0
    # identical patches give identical results up to atomics order
    if n > 8:
        d = np.abs(out[0] - out[8]).max()
        assert d < 0.05, d
    print(f"{n:4d} x {edge}^3  {dt*1e3:8.1f} ms  ok", flush=True)
print("float bm4d shim:", bm4d(synth_volume((40, 44, 48), seed=1)[0], 24.0).shape)
