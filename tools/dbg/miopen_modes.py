"""U-Net forward (batch 32 x 64^3, fp32) under MIOpen find modes / memory formats, one process per setting:
first-call seconds (solver selection) and steady-state ms.  python tools/dbg/miopen_modes.py <find_mode|-> <ndhwc 0|1> <benchmark 0|1>"""
import os
import sys
import time

mode, ndhwc, bench_flag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if mode != "-":
    os.environ["MIOPEN_FIND_MODE"] = mode
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
import torch  # noqa: E402
from aind_exaspim_image_compression.machine_learning import unet3d  # noqa: E402

torch.manual_seed(0)
torch.backends.cudnn.benchmark = bool(bench_flag)
model = unet3d.UNet().cuda().eval()
x = torch.randn(32, 1, 64, 64, 64, device="cuda")
if ndhwc:
    model = model.to(memory_format=torch.channels_last_3d)
    x = x.contiguous(memory_format=torch.channels_last_3d)
with torch.no_grad():
    t0 = time.perf_counter()
    model(x)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
print(f"find_mode {mode} ndhwc {ndhwc} benchmark {bench_flag}: first call {first:.1f} s, steady {ms:.1f} ms "
      f"({32 * 109.639e9 / ms / 1e9:.1f} TFLOP/s)", flush=True)
