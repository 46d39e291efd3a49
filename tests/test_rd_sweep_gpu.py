"""BASELINE.json config 5's rate-distortion sweep as a measurement (tools/rd_sweep.py) at 256^3: with
REAL coded bytes the rate falls and the distortion grows monotonically along both axes -- BM4D sigma
(lossless leg) and quantiser step (lossy leg) -- and the lossless leg is smaller than byte shuffle +
zstd-5 on the same chunks wherever there is denoising (sigma >= 16)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_rate_falls_and_distortion_grows_with_real_bytes():
    import rd_sweep
    res = rd_sweep.sweep(256, [0.0, 8.0, 16.0, 24.0, 32.0], [1.0, 2.0, 4.0, 8.0, 16.0, 32.0], zstd_chunks=64)
    rows = res["rows"]
    assert [r["sigma"] for r in rows] == [0.0, 8.0, 16.0, 24.0, 32.0]
    for a, b in zip(rows, rows[1:]):                      # stronger denoising: fewer bytes, further from the input
        assert b["lossless_bytes"] < a["lossless_bytes"]
        assert b["mae_vs_noisy"] > a["mae_vs_noisy"] and b["ssim_vs_noisy"] < a["ssim_vs_noisy"]
    assert rows[0]["mae_vs_noisy"] == 0.0 and rows[0]["cratio"] < 2.5 < rows[3]["cratio"]
    # the matched sigma (24 = the noise) is the best estimate of the clean volume among the lossless points
    best = max(rows, key=lambda r: r["psnr_vs_clean_db"])
    assert best["sigma"] in (24.0, 32.0) and best["psnr_vs_clean_db"] > rows[0]["psnr_vs_clean_db"] + 5.0
    for r in rows:
        for a, b in zip(r["dct"], r["dct"][1:]):          # coarser step: fewer bytes, larger error
            assert b["coded_bytes"] < a["coded_bytes"] and b["mae"] > a["mae"] and b["lmax"] >= a["lmax"]
            assert b["psnr_vs_denoised_db"] < a["psnr_vs_denoised_db"]
        assert all(p["bits_per_voxel"] < p["order0_bits_per_voxel"] for p in r["dct"] if p["q"] >= 2.0)
        if "cratio_zstd5_shuffle_sampled" in r and r["sigma"] >= 16.0:
            assert r["cratio_same_chunks"] > r["cratio_zstd5_shuffle_sampled"]
