"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

Importable subset of the reference used here: transforms, inference, unet3d,
machine_learning/metrics (numpy/scipy only).  utils/img_util (ssim3D, compute_mae) is NOT importable
here -- it imports cloudvolume, tensorstore, zarr ... at module level -- so SSIM has no fixture.
The reference's BM4D is a third-party wheel that is not installed, so nothing here covers BM4D
(parity unpinned for that part).

The fixtures are data only: inputs are regenerated from seeds, expected outputs are stored.
Host recorded in meta.json because numpy's fp32 arcsinh/sinh depend on the SIMD ISA.
"""
import json
import os
import platform
import sys

import numpy as np
import torch

from aind_exaspim_image_compression import inference as ref_inf
from aind_exaspim_image_compression.machine_learning import metrics as ref_metrics
from aind_exaspim_image_compression.machine_learning import transforms as ref_tf
from aind_exaspim_image_compression.machine_learning import unet3d as ref_unet

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from util import metric_inputs  # noqa: E402  (tests/util.py: the seeded inputs the tests regenerate)

TRANSFORM_CFGS = {
    "asinh_s32": {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}},
    "asinh_s32_o35": {"kind": "asinh", "params": {"offset": 35.0, "scale": 32.0}},
    "anscombe_g8_rn5_o100": {"kind": "anscombe",
                             "params": {"gain": 8.0, "read_noise": 5.0, "offset": 100.0}},
    "anscombe_g8_rn5_o100_alg": {"kind": "anscombe",
                                 "params": {"gain": 8.0, "read_noise": 5.0, "offset": 100.0,
                                            "unbiased_inverse": False}},
    "linear_35_1000_8": {"kind": "linear", "params": {"mn": 35.0, "mx": 1000.0, "clip": 8.0}},
    "offset37_asinh_s32": {"kind": "offset",
                           "base": {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}},
                           "params": {"offset": 37.0}},
    "offset120_anscombe": {"kind": "offset",
                           "base": {"kind": "anscombe",
                                    "params": {"gain": 8.0, "read_noise": 5.0}},
                           "params": {"offset": 120.0}},
}


def transform_inputs():
    u16 = np.arange(0, 65536, 3, dtype=np.uint16)
    grid = np.linspace(-0.05, 1.05, 20001, dtype=np.float32)
    return u16, grid


def make_transforms():
    u16, grid = transform_inputs()
    out = {}
    for name, cfg in TRANSFORM_CFGS.items():
        t = ref_tf.build_transform(cfg)
        out[f"{name}/forward_u16"] = t.forward(u16)
        out[f"{name}/inverse_u16"] = t.inverse(grid)
        out[f"{name}/inverse_float"] = np.asarray(t.inverse_float(grid), dtype=np.float32)
        out[f"{name}/norm"] = np.float64(getattr(t, "_norm", np.nan))
    # the exact vectors of reference tests/test_transforms.py:26,62,104,141,164
    t = ref_tf.AsinhTransform(offset=35, scale=32)
    v = np.array([0, 100, 1000, 10000, 60000, 65535], dtype=np.float32)
    out["tt26/rec"] = t.inverse(t.forward(v))
    t = ref_tf.AnscombeTransform(gain=8, read_noise=5, offset=100, unbiased_inverse=False)
    v = np.array([100, 500, 2000, 20000, 65535], dtype=np.float32)
    out["tt62/rec"] = t.inverse(t.forward(v))
    t = ref_tf.LinearClipTransform(mn=35, mx=1000, clip=8)
    v = np.array([35, 200, 1000, 5000], dtype=np.float32)
    out["tt104/rec"] = t.inverse(t.forward(v))
    base = ref_tf.build_transform({"kind": "asinh", "params": {"scale": 32}})
    sh = ref_tf.with_offset(base, 120.0)
    out["tt141/fwd"] = sh.forward(np.array([120.0, 152.0, 1120.0, 60120.0]))
    base = ref_tf.build_transform({"kind": "anscombe", "params": {"gain": 8, "read_noise": 5}})
    sh = ref_tf.with_offset(base, 120.0)
    out["tt164/fwd"] = sh.forward(np.array([120.0, 500.0, 2000.0, 20000.0]))
    # estimate_offset (transforms.py:414-438)
    rng = np.random.default_rng(7)
    sample = rng.integers(0, 400, size=5000).astype(np.uint16)
    sample[::13] = 0
    out["estimate_offset/p1"] = np.float64(ref_tf.estimate_offset(sample, percentile=1.0))
    out["estimate_offset/p0.1"] = np.float64(ref_tf.estimate_offset(sample, percentile=0.1))
    out["estimate_offset/p5_keepzeros"] = np.float64(
        ref_tf.estimate_offset(sample, percentile=5.0, ignore_zeros=False))
    np.savez_compressed(os.path.join(HERE, "transforms.npz"), **out)


class _Identity(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return x


class _Affine(_Identity):
    def forward(self, x):
        return x * 0.5 + 0.125


class _Shape:
    """count_patches / generate_patch_starts only read ``.shape``."""

    def __init__(self, shape):
        self.shape = shape


TILING_CASES = {
    # name: (shape, patch, overlap, trim, batch)
    "default_80x70x66": ((80, 70, 66), 64, 12, 5, 32),
    "small_50x41x37": ((50, 41, 37), 32, 6, 3, 5),
    "exact_64": ((64, 64, 64), 64, 12, 5, 32),
}


def tiling_volume(shape, seed=0):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 3000, size=shape).astype(np.uint16)


def make_tiling():
    out = {}
    shapes = [64, 65, 100, 116, 117, 256, 1024]
    for s in shapes:
        out[f"count/{s}"] = np.int64(ref_inf.count_patches(_Shape((1, 1, s, s, s)), 64, 12))
        starts = [st[0] for st in ref_inf.generate_patch_starts(_Shape((1, 1, s, 64, 64)), 64, 12)]
        out[f"starts_axis/{s}"] = np.array(starts, dtype=np.int64)
    tcfg = {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}}
    base = ref_tf.build_transform(tcfg)
    tf = ref_tf.with_offset(base, 37.0)
    for name, (shape, patch, overlap, trim, batch) in TILING_CASES.items():
        vol = tiling_volume(shape)
        for mname, model in (("identity", _Identity()), ("affine", _Affine())):
            pred = ref_inf.predict(vol, model, tf, batch_size=batch, patch_size=patch,
                                   overlap=overlap, trim=trim, verbose=False)
            out[f"predict/{name}/{mname}"] = pred
    np.savez_compressed(os.path.join(HERE, "tiling.npz"), **out)


def make_unet():
    torch.manual_seed(0)
    model = ref_unet.UNet()
    model.eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 1, 32, 32, 32, generator=g)
    torch.set_num_threads(8)
    with torch.no_grad():
        y = model(x)
    sums = {k: float(v.double().abs().sum()) for k, v in model.state_dict().items()}
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "unet.npz"), y=y.numpy())
    with open(os.path.join(HERE, "unet_state.json"), "w") as f:
        json.dump({"abs_sums": sums, "shapes": shapes,
                   "n_params": int(sum(p.numel() for p in model.parameters()))}, f, indent=1)


def make_metrics():
    """Reference outputs of machine_learning/metrics.py:306-424 and transforms.estimate_offset
    for seeded examples (inputs are regenerated from the seed by metric_inputs)."""
    out = {}
    for seed in (0, 1):
        pu, pf, raw, target, fg = metric_inputs(seed)
        for pname, pred in (("u16", pu), ("f32", pf)):
            for rname, r in (("u16", raw), ("f32", raw.astype(np.float32))):
                tag = f"s{seed}/{pname}_{rname}"
                ev = ref_metrics.evaluate_example(pred, r, target, fg)
                for k, v in ev.items():
                    out[f"{tag}/evaluate/{k}"] = np.float64(v)
                out[f"{tag}/fb_mae"] = np.array(
                    ref_metrics.foreground_background_mae(pred, r, fg), dtype=np.float64)
                out[f"{tag}/mip_max_error"] = np.float64(ref_metrics.mip_max_error(pred, r))
                out[f"{tag}/false_bright_k3"] = np.float64(
                    ref_metrics.false_bright_rate(pred, r, fg, k=3.0))
        for pct in (0.0, 0.1, 1.0, 50.0, 99.9, 100.0):
            out[f"s{seed}/estimate_offset/u16/{pct}"] = np.float64(
                ref_tf.estimate_offset(raw, percentile=pct))
            out[f"s{seed}/estimate_offset/u16_keepzeros/{pct}"] = np.float64(
                ref_tf.estimate_offset(raw, percentile=pct, ignore_zeros=False))
            out[f"s{seed}/estimate_offset/f32/{pct}"] = np.float64(
                ref_tf.estimate_offset(pf - 125.0, percentile=pct))
    # reference tests/test_metrics.py:115-138 and tests/test_transforms.py:120-129 known answers
    out["kat/fb_mae"] = np.array(ref_metrics.foreground_background_mae(
        np.array([[10.0, 20.0]]), np.array([[0.0, 0.0]]), np.array([[True, False]])))
    out["kat/mip"] = np.float64(ref_metrics.mip_max_error(np.array([1.0, 900.0]),
                                                          np.array([0.0, 1000.0])))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["metrics"]:
        make_metrics()
        print("metrics.npz written")
        sys.exit(0)
    make_transforms()
    make_metrics()
    make_tiling()
    make_unet()
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump({"numpy": np.__version__, "torch": torch.__version__,
                   "machine": platform.machine(), "processor": platform.processor(),
                   "cpu_features": "AVX512_SKX (numpy dispatches SVML fp32 arcsinh/sinh)",
                   "reference": "AllenNeuralDynamics/aind-exaspim-image-compression @ /root/reference"},
                  f, indent=1)
    print("golden fixtures written to", HERE)
