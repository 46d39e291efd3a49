"""Does MIOPEN_FIND_MODE still take effect when it is set after `import torch` (but before the first convolution)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
import torch
t = torch.zeros(4, device="cuda") + 1          # HIP is up, MIOpen has not been asked for anything yet
os.environ.setdefault("MIOPEN_FIND_MODE", sys.argv[1])
from aind_exaspim_image_compression.machine_learning import unet3d
torch.manual_seed(0)
model = unet3d.UNet().cuda().eval()
x = torch.randn(32, 1, 64, 64, 64, device="cuda")
with torch.no_grad():
    t0 = time.perf_counter(); model(x); torch.cuda.synchronize(); first = time.perf_counter() - t0
    model(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): model(x)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
print(f"late env MIOPEN_FIND_MODE={sys.argv[1]}: first call {first:.1f} s, steady {ms:.1f} ms", flush=True)
