"""Patch-cache writer and readers.

The reference precomputes, per training / validation patch, the offset-subtracted raw counts, the
clipped BM4D teacher and the foreground mask (``scripts/precompute.py:126-240``) into

    raw.npy      float32 (N, *patch_shape)     teacher.npy  float32 (N, *patch_shape)
    fg.npy       uint8   (N, *patch_shape)     transform.json / config.json

and trains from those files alone (``machine_learning/data_handling.py:1015-1217``,
``scripts/train_bm4dnet.py:14-79``).  BM4D is what makes that precompute expensive (one CPU process
per 64^3 patch); here the teacher of a whole batch of patches is one call into the HIP path
(``bm4d.denoise_patches``), and the files written are the reference's, byte-layout included, so
its ``CachedPatchDataset`` reads a cache written here and the readers below read a cache written
by the reference.

Provided with the reference's names and behaviour: ``build_training_example`` (:44-82),
``CachedPatchDataset`` (:1015-1187), ``CachedValidateDataset`` (:1190-1217); plus
``PatchCacheWriter`` / ``write_patch_cache`` (the file-writing half of ``precompute()``) and
``load_cached_transform`` (``scripts/train_bm4dnet.py:42-79``).  The cloud datasets, samplers and
mask builders of the reference module are out of scope: the writer takes the
patches and masks from the caller.
"""
import json
import os
from collections.abc import Iterable

import numpy as np
from numpy.lib.format import open_memmap

from aind_exaspim_image_compression.machine_learning.transforms import build_transform

try:  # the readers are torch Datasets when torch is there, plain sequences otherwise
    from torch.utils.data import Dataset
except ImportError:  # pragma: no cover
    class Dataset:
        pass

REQUIRED_CACHE_FILES = ("raw.npy", "teacher.npy", "fg.npy", "transform.json")
SEED_STREAMS = {"train": 0, "val": 1}          # scripts/precompute.py:60
COUNT_DTYPE = np.float32                       # scripts/precompute.py:67

# every key the reference stamps into config.json (scripts/precompute.py:172-202)
CONFIG_KEYS = (
    "split", "cache_dir", "n_patches", "brain_ids_path", "img_prefixes_path",
    "segmentation_prefixes_path", "offsets_path", "swc_pointers", "transform_cfg",
    "foreground_sampling_rate", "min_foreground_voxels", "min_segmentation_volume", "patch_shape",
    "skeleton_radius", "segmentation_dilate", "sigma_bm4d", "reject_incoherent_patches",
    "coherence_min_autocorr", "coherence_max_highfreq_frac", "coherence_min_segment_voxels",
    "coherence_smooth_sigma", "coherence_lag", "max_resample_attempts", "seed", "seed_stream",
    "num_workers", "count_dtype",
)


def build_training_example(transform, preserve_foreground, raw, teacher, fg_mask):
    """(x, y, fg_mask) from count-space arrays: the target keeps the raw counts on the foreground
    when ``preserve_foreground``, both go through ``transform.forward`` (HIP kernel), the mask
    comes back as float32 0/1."""
    fg = np.asarray(fg_mask).astype(bool)
    target = np.where(fg, raw, teacher) if preserve_foreground else teacher
    return transform.forward(raw), transform.forward(target), fg.astype(np.float32)


# ---- readers --------------------------------------------------------------------------------------------
class CachedPatchDataset(Dataset):
    """Reads precomputed count-space patches (memory-mapped) from one or several cache
    directories and applies the transform + target construction per item."""

    def __init__(self, cache_dir, transform=None, preserve_foreground=True):
        super().__init__()
        if isinstance(cache_dir, (str, os.PathLike)):
            cache_dirs = [cache_dir]
        elif isinstance(cache_dir, Iterable):
            cache_dirs = list(cache_dir)
        else:
            raise TypeError("cache_dir must be a path or an iterable of paths")
        self.raw = self._load_cached_arrs(cache_dirs, "raw")
        self.teacher = self._load_cached_arrs(cache_dirs, "teacher")
        self.fg = self._load_cached_arrs(cache_dirs, "fg")
        self._validate_cache()
        self.lengths = [len(x) for x in self.raw]
        self.cumulative_lengths = np.cumsum(self.lengths)
        self.transform = transform or build_transform({"kind": "asinh"})
        self.preserve_foreground = preserve_foreground
        self.patch_shape = tuple(self.raw[0].shape[1:])

    def __len__(self):
        return int(self.cumulative_lengths[-1])

    def __getitem__(self, idx):
        raw, teacher, fg_mask = self._get_arrays(idx)
        return build_training_example(self.transform, self.preserve_foreground, raw, teacher,
                                      fg_mask)

    def _get_arrays(self, idx):
        cache_idx, local = self._locate(idx)
        raw = np.asarray(self.raw[cache_idx][local], dtype=np.float32)
        teacher = np.asarray(self.teacher[cache_idx][local], dtype=np.float32)
        fg_mask = np.asarray(self.fg[cache_idx][local], dtype=np.float32)
        return raw, teacher, fg_mask

    def _locate(self, idx):
        if idx < 0 or idx >= len(self):
            raise IndexError(idx)
        cache_idx = int(np.searchsorted(self.cumulative_lengths, idx, side="right"))
        offset = idx if cache_idx == 0 else idx - self.cumulative_lengths[cache_idx - 1]
        return cache_idx, int(offset)

    @staticmethod
    def _load_cached_arrs(cache_dirs, name):
        return [np.load(os.path.join(d, f"{name}.npy"), mmap_mode="r") for d in cache_dirs]

    def _validate_cache(self):
        assert len(self.raw) == len(self.teacher) == len(self.fg)
        if len(self.raw) == 0:
            raise ValueError("No cached arrays found")
        for raw, teacher, fg in zip(self.raw, self.teacher, self.fg):
            assert len(raw) == len(teacher) == len(fg)
            assert raw.shape[1:] == teacher.shape[1:] == fg.shape[1:]
        shapes = {r.shape[1:] for r in self.raw}
        if len(shapes) > 1:
            raise ValueError(f"Inconsistent patch shapes across cache_dirs: {shapes}")


class CachedValidateDataset(CachedPatchDataset):
    """Cached validation examples: ``(x, y, raw, fg_mask)`` -- the raw counts ride along for the
    count-space metrics."""

    def __getitem__(self, idx):
        raw, teacher, fg_mask = self._get_arrays(idx)
        x, y, fg = build_training_example(self.transform, self.preserve_foreground, raw, teacher,
                                          fg_mask)
        return x, y, raw, fg


def _normalize_cache_dirs(cache_dir, name):
    if cache_dir is None:
        raise ValueError(f"{name} is required for training")
    if isinstance(cache_dir, (str, os.PathLike)):
        dirs = [os.fspath(cache_dir)]
    else:
        dirs = [os.fspath(d) for d in cache_dir]
    if not dirs:
        raise ValueError(f"{name} is required for training")
    return dirs


def load_cached_transform(train_cache_dir, val_cache_dir):
    """Validate every train / validation cache (directory exists, the four required files are
    there, all ``transform.json`` agree) and return their shared transform -- the contract of
    ``scripts/train_bm4dnet.py:42-79``, error types and messages included."""
    groups = {"train_cache_dir": _normalize_cache_dirs(train_cache_dir, "train_cache_dir"),
              "val_cache_dir": _normalize_cache_dirs(val_cache_dir, "val_cache_dir")}
    transform_cfg = None
    for name, dirs in groups.items():
        for index, cache_dir in enumerate(dirs):
            label = name if len(dirs) == 1 else f"{name}[{index}]"
            if not os.path.isdir(cache_dir):
                raise FileNotFoundError(f"{label} does not exist or is not a directory: {cache_dir}")
            missing = [f for f in REQUIRED_CACHE_FILES
                       if not os.path.isfile(os.path.join(cache_dir, f))]
            if missing:
                raise FileNotFoundError(f"{label} is missing required cache files: "
                                        + ", ".join(missing))
            with open(os.path.join(cache_dir, "transform.json")) as f:
                current = json.load(f)
            if transform_cfg is None:
                transform_cfg = current
            elif current != transform_cfg:
                raise ValueError("train and validation patch caches use different transforms: "
                                 f"{label}")
    return build_transform(transform_cfg)


# ---- writer ---------------------------------------------------------------------------------------------
def _write_json(path, obj):
    with open(path, "w") as f:
        json.dump(obj, f)


class PatchCacheWriter:
    """Streams (raw counts, foreground mask) patches into a reference-format cache, computing the
    BM4D teacher on the GPU a batch of patches at a time.

    ``transform_cfg`` is resolved through ``build_transform`` and stamped as ``transform.json``
    when the writer is closed; ``config.json`` is written first and carries every key the
    reference records (``CONFIG_KEYS``) -- settings that only concern the reference's cloud
    samplers can be passed through ``config`` and default to ``None``.  Offset calibration is
    refused like the reference refuses it (``precompute.py:128-137``): the cached counts must
    already have their offset subtracted."""

    def __init__(self, cache_dir, n_patches, patch_shape=(64, 64, 64), transform_cfg=None,
                 sigma_bm4d=24.0, split="train", seed=None, config=None, batch_patches=32):
        transform_cfg = transform_cfg or {"kind": "asinh"}
        if transform_cfg.get("calibrate", {}).get("offset", False):
            raise ValueError("offset calibration is not supported by the cached path; bake the "
                             "offset into transform_cfg or use per-brain offsets")
        if split not in SEED_STREAMS:
            raise ValueError(f"split must be one of {sorted(SEED_STREAMS)}")
        self.cache_dir = os.fspath(cache_dir)
        self.n_patches = int(n_patches)
        self.patch_shape = tuple(int(s) for s in patch_shape)
        self.transform = build_transform(transform_cfg)
        self.sigma_bm4d = sigma_bm4d
        self.batch_patches = int(batch_patches)
        os.makedirs(self.cache_dir, exist_ok=True)
        cfg = {k: None for k in CONFIG_KEYS}
        cfg.update(config or {})
        cfg.update({
            "split": split, "cache_dir": self.cache_dir, "n_patches": self.n_patches,
            "transform_cfg": self.transform.cfg, "patch_shape": self.patch_shape,
            "sigma_bm4d": sigma_bm4d, "seed": seed, "seed_stream": SEED_STREAMS[split],
            "count_dtype": np.dtype(COUNT_DTYPE).name,
        })
        unknown = set(cfg) - set(CONFIG_KEYS)
        if unknown:
            raise ValueError(f"unknown config keys: {sorted(unknown)}")
        _write_json(os.path.join(self.cache_dir, "config.json"), cfg)
        shape = (self.n_patches,) + self.patch_shape
        self.raw = open_memmap(os.path.join(self.cache_dir, "raw.npy"), mode="w+",
                               dtype=COUNT_DTYPE, shape=shape)
        self.teacher = open_memmap(os.path.join(self.cache_dir, "teacher.npy"), mode="w+",
                                   dtype=COUNT_DTYPE, shape=shape)
        self.fg = open_memmap(os.path.join(self.cache_dir, "fg.npy"), mode="w+", dtype=np.uint8,
                              shape=shape)
        self.written = 0

    def write(self, raw, fg_mask):
        """Append a batch: ``raw`` (B, *patch_shape) offset-subtracted counts, ``fg_mask`` the
        matching boolean masks.  The teacher is ``clip(bm4d(raw, sigma), 0, max_count)``
        (``data_handling.py:332-333``) from the HIP path."""
        from aind_exaspim_image_compression.bm4d import denoise_patches
        raw = np.asarray(raw, dtype=COUNT_DTYPE)
        fg_mask = np.asarray(fg_mask)
        if raw.ndim == len(self.patch_shape):
            raw, fg_mask = raw[None], fg_mask[None]
        if raw.shape[1:] != self.patch_shape or fg_mask.shape != raw.shape:
            raise ValueError("patch / mask shape does not match the cache's patch_shape")
        if self.written + len(raw) > self.n_patches:
            raise ValueError("more patches than the cache was allocated for")
        for b0 in range(0, len(raw), self.batch_patches):
            chunk = np.ascontiguousarray(raw[b0:b0 + self.batch_patches])
            teacher = denoise_patches(chunk, sigma=float(self.sigma_bm4d),
                                      max_count=float(self.transform.max_count))
            i0 = self.written
            self.raw[i0:i0 + len(chunk)] = chunk
            self.teacher[i0:i0 + len(chunk)] = teacher
            self.fg[i0:i0 + len(chunk)] = np.asarray(fg_mask[b0:b0 + len(chunk)], dtype=np.uint8)
            self.written += len(chunk)

    def close(self):
        """Flush the arrays and stamp ``transform.json`` (last, like the reference: a cache
        without it is incomplete and ``load_cached_transform`` rejects it)."""
        if self.written != self.n_patches:
            raise ValueError(f"cache holds {self.written} of {self.n_patches} patches")
        for arr in (self.raw, self.teacher, self.fg):
            arr.flush()
        _write_json(os.path.join(self.cache_dir, "transform.json"), self.transform.cfg)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()


def write_patch_cache(cache_dir, patches, n_patches, **kwargs):
    """Write a cache from an iterable of ``(raw, fg_mask)`` patches or batches of patches."""
    with PatchCacheWriter(cache_dir, n_patches, **kwargs) as w:
        for raw, fg_mask in patches:
            w.write(raw, fg_mask)
    return w.cache_dir
