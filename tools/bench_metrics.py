"""Row f-4 kernels on a resident volume: uint16 histogram, masked error statistics, SSIM.

usage: python tools/bench_metrics.py [edge=1024]
Prints one JSON line with milliseconds per call and the HBM rate of the streaming kernels."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
sys.path.insert(0, ROOT)

from aind_exaspim_image_compression import _native  # noqa: E402
from bench import synth_u16  # noqa: E402


def timed(ctx, fn, reps=3):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    shape = (edge,) * 3
    n = edge ** 3
    ctx = _native.context(0)
    a = synth_u16(shape, 1)
    b = synth_u16(shape, 2)
    mask = (a > 200).view(np.uint8)
    d_a, d_b, d_m = ctx.to_device(a), ctx.to_device(b), ctx.to_device(mask)
    out = {"edge": edge}
    ms = timed(ctx, lambda: ctx.u16_histogram(d_a, n))
    out["u16_histogram_ms"] = ms
    out["u16_histogram_GBps"] = 2 * n / ms / 1e6
    ms = timed(ctx, lambda: ctx.masked_error_stats(d_a, np.uint16, d_b, np.uint16, d_m, n, 500.0))
    out["masked_error_stats_ms"] = ms
    out["masked_error_stats_GBps"] = 5 * n / ms / 1e6
    ms = timed(ctx, lambda: ctx.key_histogram(d_a, np.uint16, n, 0))
    out["key_histogram_ms"] = ms
    ms = timed(ctx, lambda: ctx.ssim3d_sum(d_a, d_b, np.uint16, shape, 16, 1.0, 9.0), reps=2)
    out["ssim3d_w16_ms"] = ms
    print(json.dumps(out))


if __name__ == "__main__":
    main()
