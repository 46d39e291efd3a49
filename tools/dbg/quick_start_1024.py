"""inference.quick_start on a fresh process: one predict() of a 1024^3 volume, no warm-up call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "aind-exaspim-image-compression_amd")]
import torch
from aind_exaspim_image_compression import inference
from aind_exaspim_image_compression.machine_learning import transforms as T, unet3d
import bench
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vol = bench.synth_u16((edge,) * 3, 1000)
torch.manual_seed(0)
model = unet3d.UNet().cuda().eval()
if len(sys.argv) < 3 or sys.argv[2] != "default":
    model = inference.quick_start(model)
tf = T.build_transform({"kind": "offset", "base": {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}},
                        "params": {"offset": 37.0}})
t0 = time.perf_counter()
out = inference.predict(vol, model, tf, batch_size=32, verbose=False)
dt = time.perf_counter() - t0
print(f"{'default' if len(sys.argv) > 2 and sys.argv[2] == 'default' else 'quick_start'}: predict({edge}^3) on a fresh process, "
      f"no warm-up: {dt:.1f} s, MIOPEN_FIND_MODE={os.environ.get('MIOPEN_FIND_MODE')}, out[:5]==37: {bool(np.all(out[:5] == 37))}", flush=True)
