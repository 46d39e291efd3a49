"""The default predict() path's forward pass (NDHWC shadow, shipped find-db records), five batches of 32 x 64^3,
for a kernel trace:  rocprofv3 --kernel-trace --stats -d gpurun_out/unet_trace -- python tools/dbg/unet_forward_trace.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import inference  # noqa: E402
inference._miopen_defaults()
import torch  # noqa: E402
from aind_exaspim_image_compression.machine_learning import unet3d  # noqa: E402

torch.manual_seed(0)
model = unet3d.UNet().cuda().eval()
run = inference._ndhwc_shadow(model)
x = torch.randn(32, 1, 64, 64, 64, device="cuda")
with torch.no_grad():
    run(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run(x)
    torch.cuda.synchronize()
print(f"forward {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per batch of 32", flush=True)
