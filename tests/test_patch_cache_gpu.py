"""Row f-3 on the GPU: a cache written through the HIP BM4D path holds the oracle's teacher, and
the cached datasets return what the reference's build_training_example would (numpy oracle)."""
import json

import numpy as np
import pytest

from oracle import host_oracle as H
from util import psnr, synth_volume

from aind_exaspim_image_compression.machine_learning import data_handling as D

pytestmark = pytest.mark.gpu
TCFG = {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}}


def patches(n, edge):
    raws, fgs = [], []
    for i in range(n):
        noisy, clean = synth_volume((edge,) * 3, seed=10 + i, sigma=24.0, pedestal=5.0)
        raws.append(noisy)
        fgs.append(clean > 60.0)
    return np.stack(raws), np.stack(fgs)


def test_written_cache_equals_oracle_teacher_and_reads_back(tmp_path, oracle):
    edge, n = 32, 3
    raw, fg = patches(n, edge)
    cache = D.write_patch_cache(tmp_path / "train", [(raw[:2], fg[:2]), (raw[2], fg[2])], n,
                                patch_shape=(edge,) * 3, transform_cfg=TCFG, sigma_bm4d=24,
                                seed=1, batch_patches=2)
    assert json.loads((tmp_path / "train" / "transform.json").read_text()) == TCFG
    got_raw = np.load(tmp_path / "train" / "raw.npy")
    got_teacher = np.load(tmp_path / "train" / "teacher.npy")
    got_fg = np.load(tmp_path / "train" / "fg.npy")
    np.testing.assert_array_equal(got_raw, raw)
    np.testing.assert_array_equal(got_fg, fg.astype(np.uint8))
    assert got_teacher.dtype == np.float32 and got_teacher.min() >= 0.0
    for i in range(n):     # teacher = clip(bm4d(raw, 24), 0, 65535): oracle within fp32 tolerance
        want = np.clip(oracle.bm4d(raw[i], 24.0), 0.0, 65535.0)
        assert psnr(got_teacher[i], want, 1000.0) > 80.0
    # the reference's contract check accepts the cache, the datasets read it
    val = D.write_patch_cache(tmp_path / "val", [(raw[:1], fg[:1])], 1, patch_shape=(edge,) * 3,
                              transform_cfg=TCFG, split="val")
    tf = D.load_cached_transform(cache, val)
    ds = D.CachedPatchDataset(cache, transform=tf)
    o = H.TransformOracle(TCFG)
    assert len(ds) == n
    for i in (0, 2):
        x, y, m = ds[i]
        target = np.where(fg[i], raw[i], got_teacher[i])
        np.testing.assert_array_equal(x, o.forward(raw[i]))
        np.testing.assert_array_equal(y, o.forward(target))
        np.testing.assert_array_equal(m, fg[i].astype(np.float32))
    vds = D.CachedValidateDataset([cache, val], transform=tf, preserve_foreground=False)
    assert len(vds) == n + 1
    x, y, r, m = vds[n]
    np.testing.assert_array_equal(r, raw[0])
    np.testing.assert_array_equal(y, o.forward(np.load(tmp_path / "val" / "teacher.npy")[0]))
    # same patch, different batch: the same teacher (integer aggregation sums, a unit per patch)
    np.testing.assert_array_equal(np.load(tmp_path / "val" / "teacher.npy")[0], got_teacher[0])
