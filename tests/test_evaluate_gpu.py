"""Row a-J: the reference's two evaluation loops (evaluate.py:88-112 and :172-217) assembled from
the device operators, checked against the same loops written with the CPU oracles."""
import zlib

import numpy as np
import pytest
import torch

from oracle import bm4d_oracle as O
from oracle import host_oracle as H
from util import synth_volume

from aind_exaspim_image_compression import bm4d as B
from aind_exaspim_image_compression import evaluate
from aind_exaspim_image_compression.machine_learning import transforms as T

pytestmark = pytest.mark.gpu
BASE_CFG = {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}}


class Shrink(torch.nn.Module):
    """Stands in for the network: pulls the normalised signal towards its local mean."""

    def forward(self, x):
        return 0.5 * x + 0.5 * torch.nn.functional.avg_pool3d(x, 3, stride=1, padding=1)


class Zlib:
    """``encode`` protocol of the numcodecs codec the reference passes (evaluate.py:40)."""

    def encode(self, chunk):
        return zlib.compress(np.ascontiguousarray(chunk).tobytes(), 6)


def ref_cratio(img, codec):
    img = np.ascontiguousarray(img, dtype=np.uint16)
    raw = packed = 0
    for z in range(0, img.shape[0], 64):
        for y in range(0, img.shape[1], 64):
            for x in range(0, img.shape[2], 64):
                chunk = np.ascontiguousarray(img[z:z + 64, y:y + 64, x:x + 64])
                packed += len(codec.encode(chunk))
                raw += chunk.nbytes
    return round(raw / packed, 2)


def test_compare_with_bm4d_equals_the_per_patch_loop():
    patches = np.stack([synth_volume((34, 34, 34), seed=s, as_u16=True)[0] for s in (3, 4)])
    model = Shrink().cuda()
    base = T.build_transform(BASE_CFG)
    got = evaluate.compare_with_bm4d(patches, model, base, codec=Zlib(), offset=37.0,
                                     keep_images=True)
    tf_o = H.TransformOracle(T.with_offset(base, 37.0).cfg)
    for i, patch in enumerate(patches):
        noise = patch[5:-5, 5:-5, 5:-5]
        want_gt = np.maximum(O.bm4d(noise.astype(np.float32), 10.0), 0).astype(int)
        x = torch.from_numpy(tf_o.forward(patch))[None, None].cuda()
        with torch.no_grad():
            y = model(x)[0, 0].cpu().numpy()
        denoised = tf_o.inverse(y)[5:-5, 5:-5, 5:-5]
        np.testing.assert_array_equal(got["denoised"][i], denoised)
        # ground truth: one batched BM4D call for all patches == one call per patch == the oracle
        gt_dev = got["denoised_gt"][i]
        assert gt_dev.dtype == np.int64
        alone = np.maximum(B.bm4d(noise, 10), 0).astype(int)
        np.testing.assert_array_equal(gt_dev, alone)
        np.testing.assert_array_equal(gt_dev, want_gt)
        assert got["cratio"][i] == ref_cratio(denoised, Zlib())
        assert got["cratio_noise"][i] == ref_cratio(noise, Zlib())
        assert got["cratio_gt"][i] == ref_cratio(gt_dev, Zlib())
        assert got["ssim_noise"][i] == pytest.approx(H.ssim3d(noise, denoised), rel=1e-12)
        assert got["ssim_gt"][i] == pytest.approx(H.ssim3d(gt_dev, denoised), rel=1e-12)
        assert got["l1_gt"][i] == pytest.approx(H.compute_mae(gt_dev, denoised), rel=1e-12)
        assert got["lmax_gt"][i] == H.compute_lmax(gt_dev, denoised)
    proxy = evaluate.compare_with_bm4d(patches[:1], model, base, offset=37.0)
    assert set(proxy) == {"cratio", "cratio_noise", "cratio_gt", "ssim_noise", "ssim_gt", "l1_gt",
                          "lmax_gt"}
    assert proxy["cratio"][0] > 1.0 and proxy["cratio_gt"][0] > proxy["cratio_noise"][0]


def test_evaluate_blocks_equals_the_per_block_loop():
    vol = synth_volume((70, 66, 80), seed=9, as_u16=True)[0]
    model = Shrink().cuda()
    base = T.build_transform(BASE_CFG)
    got = evaluate.evaluate_blocks({"block_001": vol[None, None], "block_000": vol[:64]}, model,
                                   base, codec=Zlib(), batch_size=4)
    assert list(got) == ["block_000", "block_001"]
    off = H.estimate_offset(vol, 0.1)
    tf_o = H.TransformOracle(T.with_offset(base, off).cfg)

    def net(batch):
        with torch.no_grad():
            return model(torch.from_numpy(batch[:, None]).cuda())[:, 0].cpu().numpy()

    denoised = H.predict(vol, net, tf_o, batch_size=4, patch=64, overlap=12, trim=5)
    row = got["block_001"]
    assert row["cratio"] == ref_cratio(denoised, Zlib())
    assert row["cratio_noise"] == ref_cratio(vol, Zlib())
    assert row["ssim"] == pytest.approx(H.ssim3d(vol, denoised, data_range=np.max(vol)), rel=1e-12)
