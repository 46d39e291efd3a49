"""PCIe-inclusive rate of the chunk-local mode on a host volume (never bench.py's `value`):

    python tools/bench_streamed.py [layers=4] [rows=2048] [cols=2048] [chunk=256] [streamed-only]

(a) exabm4d_denoise_chunked_u16_host -- layers of chunks streamed, copies under the kernels;
(b) the one-call form: whole volume up, exabm4d_denoise_chunked_u16_dev, whole volume down;
(c) the device call of (b) alone.  The volume is `layers` copies of one synthetic layer (bench.synth_u16)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "aind-exaspim-image-compression_amd")]
import bench  # noqa: E402
from aind_exaspim_image_compression import _native  # noqa: E402
from aind_exaspim_image_compression.bm4d import denoise_chunked_streamed  # noqa: E402

only = "streamed-only" in sys.argv
argv = [a for a in sys.argv[1:] if a != "streamed-only"]
layers, rows, cols, chunk = (int(v) for v in (argv[:4] + ["4", "2048", "2048", "256"][len(argv):]))
halo = 8
base = bench.synth_u16((chunk, rows, cols), seed=4000)
vol = np.concatenate([base] * layers)
del base
out = np.empty_like(vol)
out[:] = 0                                            # touch the pages: the timed calls do not pay for first use
ctx = _native.context(0)
nvox = vol.size
res = {"volume": list(vol.shape), "chunk": chunk, "halo": halo, "GB_each_way": vol.nbytes / 1e9}

denoise_chunked_streamed(vol[:chunk], bench.SIGMA, bench.OFFSET, chunk=chunk, halo=halo, out=out[:chunk])  # warm-up
t0 = time.perf_counter()
denoise_chunked_streamed(vol, bench.SIGMA, bench.OFFSET, chunk=chunk, halo=halo, out=out)
ts = time.perf_counter() - t0
res["streamed_s"] = ts
res["streamed_voxels_per_s"] = nvox / ts
res["residual_std"] = float((out[::4, ::8, ::8].astype(np.float32) - vol[::4, ::8, ::8].astype(np.float32)).std())
if only:
    print(json.dumps(res))
    sys.exit(0)

t0 = time.perf_counter()
d_in = ctx.to_device(vol)
d_out = ctx.alloc(vol.nbytes)
t1 = time.perf_counter()
ctx.denoise_chunked_u16(d_in, d_out, vol.shape, bench.SIGMA, bench.OFFSET, chunk=chunk, halo=halo)
ctx.sync()
t2 = time.perf_counter()
one = d_out.download(vol.shape, np.uint16)
t3 = time.perf_counter()
res["one_call_s"] = t3 - t0
res["one_call_voxels_per_s"] = nvox / (t3 - t0)
res["one_call_parts_s"] = {"upload": t1 - t0, "device": t2 - t1, "download": t3 - t2}
res["device_only_voxels_per_s"] = nvox / (t2 - t1)
d = np.abs(one[::3, ::5, ::7].astype(np.int32) - out[::3, ::5, ::7].astype(np.int32))
res["streamed_vs_one_call"] = {"max_abs": int(d.max()), "frac_differing": float(np.mean(d > 0))}
print(json.dumps(res))
