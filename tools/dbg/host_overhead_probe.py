"""Where the host-to-host time of a batched denoise_patches call goes (1000 patches of 64^3 fp32):
python tools/dbg/host_overhead_probe.py [n]"""
import ctypes, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "aind-exaspim-image-compression_amd"), os.path.join(R, "tests")]
from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.bm4d import denoise_patches
from util import synth_volume
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = _native.context(0)
base = np.stack([synth_volume((64,) * 3, seed=i)[0] for i in range(8)])
raw = np.concatenate([base] * ((n + 7) // 8))[:n].copy()
p = _native.default_params()
lib = _native.lib()
def call(out):
    t0 = time.perf_counter()
    rc = lib.exabm4d_denoise_f32_host(ctx.handle, raw.ctypes.data, out.ctypes.data, 64, 64, 64, n, 24.0, ctypes.byref(p), 2, 0.0, 65535.0)
    assert rc == 0
    return (time.perf_counter() - t0) * 1e3
denoise_patches(raw[:32], 24.0)
for rep in range(4):
    ctx.set_option("host_pipeline", rep & 1)
    print("host_pipeline", rep & 1)
    t0 = time.perf_counter(); out = denoise_patches(raw, 24.0); dt = (time.perf_counter() - t0) * 1e3
    print(f"denoise_patches({n} x 64^3): {dt:7.1f} ms", flush=True)
    fresh = np.empty_like(raw)
    print(f"C entry, fresh destination:      {call(fresh):7.1f} ms", flush=True)
    print(f"C entry, same destination again: {call(fresh):7.1f} ms", flush=True)
    t0 = time.perf_counter(); z = np.empty_like(raw); z.fill(1); dt = (time.perf_counter() - t0) * 1e3
    print(f"np.empty + fill of {raw.nbytes >> 20} MiB:      {dt:7.1f} ms", flush=True)
    d_in, d_out = ctx.to_device(raw), ctx.alloc(raw.nbytes)
    ev = [ctx.event(), ctx.event()]
    ctx.record(ev[0]); ctx.denoise_f32(d_in, d_out, (64, 64, 64), 24.0, batch=n, clip=(0.0, 65535.0)); ctx.record(ev[1]); ctx.sync()
    print(f"on the device:                   {ctx.elapsed_ms(ev[0], ev[1]):7.1f} ms", flush=True)
    d_in.free(); d_out.free()
