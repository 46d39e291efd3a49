"""U-Net forward (32 x 64^3, fp32) under MIOpen settings: default, exhaustive find (cudnn.benchmark),
channels_last_3d.  Prints ms per forward and the max difference to the default result."""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression.machine_learning import unet3d

torch.manual_seed(0)
model = unet3d.UNet().cuda().eval()
x = torch.randn(32, 1, 64, 64, 64, device="cuda")


def run(tag, m, inp):
    with torch.no_grad():
        for _ in range(2):
            y = m(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = m(inp)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{tag:32s} {dt*1e3:8.1f} ms  {32*109.64e9/dt/1e12:6.1f} TFLOP/s", flush=True)
    return y.float().contiguous()


y0 = run("default", model, x)
torch.backends.cudnn.benchmark = True
y1 = run("cudnn.benchmark", model, x)
print("   max |diff| vs default:", float((y1 - y0).abs().max()), flush=True)
m2 = model.to(memory_format=torch.channels_last_3d)
y2 = run("channels_last_3d + benchmark", m2, x.to(memory_format=torch.channels_last_3d))
print("   max |diff| vs default:", float((y2 - y0).abs().max()), flush=True)
