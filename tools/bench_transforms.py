"""HBM-roofline check of the fused intensity-transform kernels (1024^3 voxels, device resident)."""
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
from aind_exaspim_image_compression import _native  # noqa: E402
from aind_exaspim_image_compression.machine_learning import transforms as T  # noqa: E402

ctx = _native.context(0)
n = 1024 ** 3
d_u16 = ctx.alloc(2 * n).zero()
d_f32 = ctx.alloc(4 * n)
d_out = ctx.alloc(2 * n)
for name, cfg in (("asinh+offset", {"kind": "offset", "base": {"kind": "asinh", "params": {"scale": 32.0}},
                                   "params": {"offset": 37.0}}),
                  ("anscombe", {"kind": "anscombe", "params": {"gain": 8.0, "read_noise": 5.0}}),
                  ("linear", {"kind": "linear", "params": {"mn": 35.0, "mx": 1000.0}})):
    tf = T.build_transform(cfg)
    for label, fn, nbytes in (
            ("forward u16->f32", lambda: tf.forward_device(ctx, d_u16, d_f32, n, True), 6 * n),
            ("inverse f32->u16", lambda: tf.inverse_device(ctx, d_f32, d_out, n, True), 6 * n)):
        fn()
        ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(3):
            fn()
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 3
        print(f"{name:14s} {label}: {ms:7.2f} ms  {nbytes / ms / 1e6:7.0f} GB/s "
              f"({nbytes / ms / 1e6 / 8000:.2f} of 8 TB/s)", flush=True)
