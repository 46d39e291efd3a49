"""Rate-distortion sweep of the denoise -> quantise -> (rate proxy) path on one MI355X
(BASELINE.json config 5, with the pieces this repo has): for each BM4D sigma the synthetic uint16
volume is denoised on the device, the rate is the order-0 entropy bound of the byte-shuffled
64^3 chunks (`shuffled_entropy_cratio`, row f-1's proxy -- NOT Blosc-zstd bytes) and the
distortion is SSIM / MAE against the noisy input (the reference's own report, evaluate.py:105:
ssim3D(noise, denoised)), all reduced on the GPU (row f-4).

A second sweep quantises the sigma = 24 result with the block-DCT quantiser of row f-1 at several
steps: order-0 entropy of the indices in bits per voxel against MAE / max error / SSIM.

usage: python tools/rd_sweep.py [edge=512] [sigmas=0,8,16,24,32,48]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
sys.path.insert(0, ROOT)

from aind_exaspim_image_compression.bm4d import denoise_volume  # noqa: E402
from aind_exaspim_image_compression.utils import dct_quant, img_util  # noqa: E402
import bench  # noqa: E402


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    sigmas = [float(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "0,8,16,24,32,48").split(",")]
    shape = (edge,) * 3
    noisy = bench.synth_u16(shape, 1)
    rows = []
    for sigma in sigmas:
        t0 = time.perf_counter()
        den = noisy if sigma == 0 else denoise_volume(noisy, sigma, offset=bench.OFFSET)
        dt = time.perf_counter() - t0
        rows.append({
            "sigma": sigma,
            "entropy_cratio": img_util.shuffled_entropy_cratio(den),
            "mae_vs_noisy": img_util.compute_mae(den, noisy),
            "ssim_vs_noisy": float(img_util.ssim3D(noisy, den, data_range=np.max(noisy))),
            "seconds_host_to_host": round(dt, 3),
        })
        print(json.dumps(rows[-1]), flush=True)
    # second axis of the sweep: the transform quantiser on the sigma = 24 result
    den = denoise_volume(noisy, 24.0, offset=bench.OFFSET)
    qrows = []
    for q in (1.0, 2.0, 4.0, 8.0, 16.0, 32.0):
        rd = dct_quant.rate_distortion(den, q)
        rec = dct_quant.reconstruct(dct_quant.quantise(den, q), den.shape, q)
        rd["ssim_vs_denoised"] = float(img_util.ssim3D(den, rec, data_range=np.max(den)))
        qrows.append(rd)
        print(json.dumps(rd), flush=True)
    print(json.dumps({"volume": shape, "rows": rows, "dct_quantiser_on_sigma24": qrows}))


if __name__ == "__main__":
    main()
