"""Row f-1 transform quantiser on a resident 1024^3 uint16 volume: ms per call and HBM rate
(algorithmic 2 B of volume + 4 B of indices per voxel each way)."""
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
sys.path.insert(0, ROOT)

from aind_exaspim_image_compression import _native  # noqa: E402
from bench import synth_u16  # noqa: E402


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    shape = (edge,) * 3
    n = edge ** 3
    ctx = _native.context(0)
    vol = synth_u16(shape, 1)
    d_vol, d_idx, d_rec = ctx.to_device(vol), ctx.alloc(n * 4), ctx.alloc(n * 2)
    out = {"edge": edge}
    for name, fn in (("dctq_forward", lambda: ctx.dctq_forward(d_vol, shape, 8.0, d_idx)),
                     ("dctq_inverse", lambda: ctx.dctq_inverse(d_idx, shape, 8.0, d_rec))):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        ctx.sync()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        out[name + "_ms"] = ms
        out[name + "_GBps"] = 6 * n / ms / 1e6
    print(json.dumps(out))


if __name__ == "__main__":
    main()
