"""Throughput of the BM4DNet stage (PyTorch-ROCm U-Net inside the device-resident predict()).
Not part of bench.py's metric; reported in DESIGN.md for BASELINE config 3."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import inference  # noqa: E402
from aind_exaspim_image_compression.machine_learning import transforms as T  # noqa: E402
from aind_exaspim_image_compression.machine_learning import unet3d  # noqa: E402

torch.manual_seed(0)
model = unet3d.UNet().cuda().eval()
x = torch.randn(32, 1, 64, 64, 64, device="cuda")
for name, ctxm in (("fp32", torch.autocast("cuda", enabled=False)),
                   ("bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16)),
                   ("fp16 autocast", torch.autocast("cuda", dtype=torch.float16))):
    with torch.no_grad(), ctxm:
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            model(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"U-Net forward, batch 32 x 64^3, {name}: {dt*1e3:.1f} ms  "
          f"({32*109.64e9/dt/1e12:.1f} TFLOP/s, {32*54**3/dt:.3e} output voxels/s)", flush=True)

tf = T.build_transform({"kind": "offset",
                        "base": {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}},
                        "params": {"offset": 37.0}})
vol = np.random.default_rng(0).integers(0, 3000, size=(256, 256, 256)).astype(np.uint16)
inference.predict(vol[:64, :64, :64], model, tf, verbose=False)
t0 = time.perf_counter()
out = inference.predict(vol, model, tf, batch_size=32, verbose=False)
dt = time.perf_counter() - t0
print(f"predict(256^3, 125 patches, fp32): {dt:.2f} s = {vol.size/dt:.3e} voxels/s", flush=True)

ident = torch.nn.Identity().cuda()
t0 = time.perf_counter()
inference.predict(vol, ident, tf, batch_size=32, verbose=False)
dt = time.perf_counter() - t0
print(f"predict(256^3) stitching alone (identity model): {dt*1e3:.1f} ms", flush=True)

# opt-in MIOpen tuning (inference.tune_model): search cost and steady state
t0 = time.perf_counter()
inference.tune_model(model)
inference.predict(vol, model, tf, batch_size=32, verbose=False)
first = time.perf_counter() - t0
t0 = time.perf_counter()
inference.predict(vol, model, tf, batch_size=32, verbose=False)
dt = time.perf_counter() - t0
print(f"tune_model: first predict(256^3) {first:.1f} s (solver search), then {dt:.2f} s = "
      f"{vol.size/dt:.3e} voxels/s", flush=True)
