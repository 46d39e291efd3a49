"""Rate-distortion sweep of BASELINE.json config 5 -- BM4D denoise -> 8^3 block DCT quantise -> entropy
encode -- on one MI355X with REAL coded bytes (EXAC v2 streams, the same calls bench.py times) and
the codec family the reference ships (byte shuffle + zstd-5 per 64^3 chunk, libzstd via ctypes) next
to the lossless points.  The volume, its denoised versions, indices and reconstructions stay in HBM;
the host sees scalars (and, for zstd, a sample of chunks).

The reference's own report is cratio(raw), cratio(denoised) and their ratio
(scripts/evaluate_bm4dnet.py:138-145) plus ssim3D(noise, denoised) (evaluate.py:105).

    python tools/rd_sweep.py [edge=1024] [sigmas=0,8,16,24,32,48] [qs=1,2,4,8,16,32] > rd_sweep.json

Axis 1 (lossless leg): for each BM4D sigma -- coded bytes of the denoised uint16 volume, distortion
against the noisy input (MAE, SSIM) and against the clean volume (PSNR, on a 256^3 corner).
Axis 2 (lossy leg): for each sigma and each step q -- coded bytes of the int32 indices, error of the
reconstruction against the denoised volume (MAE, max) and PSNR against the clean corner."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aind-exaspim-image-compression_amd"))
sys.path.insert(0, ROOT)

from aind_exaspim_image_compression import _native  # noqa: E402
from aind_exaspim_image_compression.utils import dct_quant  # noqa: E402
import bench  # noqa: E402


def sweep(edge, sigmas, qs, seed=1000, zstd_chunks=128, log=None):
    from oracle import zstd_ref           # checker-side comparison only (tool, not product)
    shape = (edge,) * 3
    n = edge ** 3
    ctx = _native.context(0)
    noisy = bench.synth_u16(shape, seed)
    corner = min(edge, 256)
    clean = bench.synth_clean((corner,) * 3, seed)
    peak = float(clean.max() - clean.min())
    csl = (slice(0, corner),) * 3
    d_noisy, d_den, d_rec = ctx.to_device(noisy), ctx.alloc(2 * n), ctx.alloc(2 * n)
    nblk = (-(-edge // 8)) ** 3
    d_idx = ctx.alloc(4 * nblk * 512)
    nchunks = (-(-edge // 64)) ** 3
    d_sz = ctx.alloc(4 * nchunks)
    pick = list(range(0, nchunks, max(1, nchunks // zstd_chunks)))
    g = -(-edge // 64)

    def box(c):
        z, y, x = c // (g * g), (c // g) % g, c % g
        return (slice(64 * z, 64 * z + 64), slice(64 * y, 64 * y + 64), slice(64 * x, 64 * x + 64))

    rows = []
    for sigma in sigmas:
        t0 = time.perf_counter()
        if sigma > 0:
            ctx.denoise_u16(d_noisy, d_den, shape, sigma, bench.OFFSET)
            src = d_den
        else:
            src = d_noisy
        coded, _ = ctx.codec_encode(src, 2, shape, bench.CHUNK, sizes=d_sz)
        ctx.sync()
        dt = time.perf_counter() - t0
        den = src.download(shape, np.uint16)
        sizes = d_sz.download((nchunks,), np.uint32).astype(np.uint64)
        err = ctx.masked_error_stats(src, np.uint16, d_noisy, np.uint16, None, n)
        ssim = ctx.ssim3d_sum(d_noisy, src, np.uint16, shape, 16, (0.01 * float(noisy.max())) ** 2,
                              (0.03 * float(noisy.max())) ** 2) / n
        row = {"sigma": sigma, "lossless_bytes": int(coded), "cratio": 2.0 * n / coded,
               "bits_per_voxel": 8.0 * coded / n, "mae_vs_noisy": float(err[1] / n),
               "ssim_vs_noisy": float(ssim), "psnr_vs_clean_db": bench.psnr_db(den[csl], clean, peak),
               "device_seconds_denoise_plus_encode": round(dt, 3)}
        if zstd_ref.available():
            zs = sum(zstd_ref.shuffle_zstd_size(den[box(c)], 5) for c in pick)
            row["cratio_zstd5_shuffle_sampled"] = float(sum(den[box(c)].nbytes for c in pick)) / zs
            row["cratio_same_chunks"] = float(sum(den[box(c)].nbytes for c in pick)) / float(sizes[pick].sum())
        row["dct"] = []
        for q in qs:
            rd = dct_quant.rate_distortion_device(ctx, src, shape, q, d_idx=d_idx, d_rec=d_rec)
            rec = d_rec.download(shape, np.uint16)
            rd["psnr_vs_clean_db"] = bench.psnr_db(rec[csl], clean, peak)
            rd["psnr_vs_denoised_db"] = bench.psnr_db(rec[csl], den[csl], peak)
            row["dct"].append(rd)
        rows.append(row)
        if log:
            print(json.dumps(row), file=log, flush=True)
    for b in (d_noisy, d_den, d_rec, d_idx, d_sz):
        b.free()
    return {"volume": list(shape), "seed": seed, "offset": bench.OFFSET, "noise_sigma": bench.SIGMA,
            "codec": "EXAC v2 (DESIGN.md 3.11b): 64^3 chunks of the uint16 volume; chunks of 512 blocks x 8 x 64 "
                     "of the int32 indices",
            "psnr_peak": peak, "psnr_region": f"{corner}^3 corner against the clean volume",
            "zstd": (f"libzstd {zstd_ref.version()} level 5 on byte-shuffled 64^3 chunks, {len(pick)} sampled chunks"
                     if zstd_ref.available() else None),
            "rows": rows}


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    sigmas = [float(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "0,8,16,24,32,48").split(",")]
    qs = [float(s) for s in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8,16,32").split(",")]
    print(json.dumps(sweep(edge, sigmas, qs, log=sys.stderr), indent=1))


if __name__ == "__main__":
    main()
