"""Row f-3 on the CPU: the reference's on-disk cache contract (file names, dtypes, config keys,
error behaviour of scripts/train_bm4dnet.py:_load_cached_transform as its tests state it) and the
index logic of the cached datasets.  Items that run the transform / BM4D need the GPU
(tests/test_patch_cache_gpu.py)."""
import json

import numpy as np
import pytest

from aind_exaspim_image_compression.machine_learning import data_handling as D

TCFG = {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}}


def make_cache(root, name, n=3, shape=(8, 8, 8), transform=None, omit=(), seed=0):
    """A cache laid out exactly as scripts/precompute.py:204-238 writes it."""
    d = root / name
    d.mkdir()
    rng = np.random.default_rng(seed)
    arrays = {"raw.npy": rng.normal(100, 20, (n,) + shape).astype(np.float32),
              "teacher.npy": rng.normal(100, 2, (n,) + shape).astype(np.float32),
              "fg.npy": (rng.random((n,) + shape) < 0.2).astype(np.uint8)}
    for fname, arr in arrays.items():
        if fname not in omit:
            np.save(d / fname, arr)
    if "transform.json" not in omit:
        (d / "transform.json").write_text(json.dumps(transform or TCFG))
    return d, arrays


def test_reader_index_logic_over_several_caches(tmp_path):
    d0, a0 = make_cache(tmp_path, "c0", n=3, seed=0)
    d1, a1 = make_cache(tmp_path, "c1", n=2, seed=1)
    ds = D.CachedPatchDataset([str(d0), str(d1)], transform=object())
    assert len(ds) == 5 and ds.patch_shape == (8, 8, 8) and ds.lengths == [3, 2]
    assert [ds._locate(i) for i in range(5)] == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1)]
    for bad in (-1, 5):
        with pytest.raises(IndexError):
            ds._locate(bad)
    raw, teacher, fg = ds._get_arrays(3)
    np.testing.assert_array_equal(raw, a1["raw.npy"][0])
    np.testing.assert_array_equal(teacher, a1["teacher.npy"][0])
    assert fg.dtype == np.float32 and set(np.unique(fg)) <= {0.0, 1.0}
    single = D.CachedValidateDataset(str(d0), transform=object())
    assert len(single) == 3
    with pytest.raises(TypeError):
        D.CachedPatchDataset(3)
    with pytest.raises(ValueError):
        D.CachedPatchDataset([])
    dbad, _ = make_cache(tmp_path, "c2", shape=(4, 4, 4))
    with pytest.raises(ValueError, match="Inconsistent patch shapes"):
        D.CachedPatchDataset([str(d0), str(dbad)], transform=object())


class _Doubling:
    cfg = TCFG

    def forward(self, x):
        return np.asarray(x, dtype=np.float32) * 2


def test_build_training_example_semantics():
    raw = np.array([1.0, 2.0, 3.0], np.float32)
    teacher = np.array([10.0, 20.0, 30.0], np.float32)
    fg = np.array([1.0, 0.0, 1.0], np.float32)
    x, y, m = D.build_training_example(_Doubling(), True, raw, teacher, fg)
    np.testing.assert_array_equal(x, [2, 4, 6])
    np.testing.assert_array_equal(y, [2, 40, 6])          # raw kept on the foreground
    assert m.dtype == np.float32 and m.tolist() == [1.0, 0.0, 1.0]
    _, y, _ = D.build_training_example(_Doubling(), False, raw, teacher, fg)
    np.testing.assert_array_equal(y, [20, 40, 60])


def test_load_cached_transform_contract(tmp_path):
    """reference tests/test_train_bm4dnet.py:44-110."""
    with pytest.raises(ValueError, match="train_cache_dir is required"):
        D.load_cached_transform(None, "/validation")
    train, _ = make_cache(tmp_path, "train")
    with pytest.raises(ValueError, match="val_cache_dir is required"):
        D.load_cached_transform(str(train), None)
    with pytest.raises(FileNotFoundError, match="does not exist"):
        D.load_cached_transform("/missing-training-cache", "/validation")
    partial, _ = make_cache(tmp_path, "partial", omit=("teacher.npy", "transform.json"))
    val, _ = make_cache(tmp_path, "val")
    with pytest.raises(FileNotFoundError, match="teacher.npy, transform.json"):
        D.load_cached_transform(str(partial), str(val))
    other, _ = make_cache(tmp_path, "val16",
                          transform={"kind": "asinh", "params": {"offset": 0.0, "scale": 16.0}})
    train1, _ = make_cache(tmp_path, "train1")
    with pytest.raises(ValueError, match="different transforms"):
        D.load_cached_transform([str(train), str(train1)], [str(other)])
    incomplete, _ = make_cache(tmp_path, "train2", omit=("teacher.npy",))
    with pytest.raises(FileNotFoundError, match=r"train_cache_dir\[1\].*teacher.npy"):
        D.load_cached_transform([str(train), str(incomplete)], str(val))
    tf = D.load_cached_transform([str(train), str(train1)], str(val))
    assert tf.cfg == TCFG


def test_writer_refuses_what_the_reference_refuses(tmp_path):
    with pytest.raises(ValueError, match="offset calibration is not supported"):
        D.PatchCacheWriter(tmp_path / "a", 1, transform_cfg={"kind": "asinh",
                                                             "calibrate": {"offset": True}})
    with pytest.raises(ValueError, match="split"):
        D.PatchCacheWriter(tmp_path / "b", 1, split="test")
    with pytest.raises(ValueError, match="unknown config keys"):
        D.PatchCacheWriter(tmp_path / "c", 1, config={"nonsense": 1})


def test_writer_stamps_the_reference_config_before_any_patch(tmp_path):
    """scripts/precompute.py:172-202 / tests/test_precompute.py: config.json is complete before
    cache generation starts; transform.json only appears on close()."""
    settings = {"brain_ids_path": "/data/brains.txt", "img_prefixes_path": "/data/images.json",
                "offsets_path": "/data/offsets.json", "foreground_sampling_rate": 0.5,
                "min_foreground_voxels": 50, "skeleton_radius": 2, "num_workers": None}
    w = D.PatchCacheWriter(tmp_path / "cache", 12, patch_shape=(8, 8, 8), transform_cfg=TCFG,
                           sigma_bm4d=24, split="val", seed=42, config=settings)
    cfg = json.loads((tmp_path / "cache" / "config.json").read_text())
    assert set(cfg) == set(D.CONFIG_KEYS)
    for k, v in settings.items():
        assert cfg[k] == v
    assert cfg["seed_stream"] == 1 and cfg["count_dtype"] == "float32" and cfg["split"] == "val"
    assert cfg["n_patches"] == 12 and cfg["patch_shape"] == [8, 8, 8] and cfg["sigma_bm4d"] == 24
    assert cfg["transform_cfg"] == TCFG and cfg["seed"] == 42
    assert not (tmp_path / "cache" / "transform.json").exists()
    for name, dtype in (("raw", np.float32), ("teacher", np.float32), ("fg", np.uint8)):
        arr = np.load(tmp_path / "cache" / f"{name}.npy", mmap_mode="r")
        assert arr.shape == (12, 8, 8, 8) and arr.dtype == dtype
    with pytest.raises(ValueError, match="0 of 12"):
        w.close()
    with pytest.raises(ValueError, match="shape"):
        w.write(np.zeros((1, 4, 4, 4), np.float32), np.zeros((1, 4, 4, 4), bool))
