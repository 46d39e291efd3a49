"""Host-side logic of the drop-in package that needs no GPU: transform construction / cfg
handling (the reference's tests/test_transforms.py cases that do not evaluate a transform),
patch-grid helpers, checkpoints, the U-Net against the reference-generated fixture (CPU)."""
import json
import os

import numpy as np
import pytest
import torch

from aind_exaspim_image_compression import inference
from aind_exaspim_image_compression.machine_learning import transforms as T
from aind_exaspim_image_compression.machine_learning import unet3d

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_build_transform_kinds_and_errors():
    assert isinstance(T.build_transform({"kind": "asinh"}), T.AsinhTransform)
    t = T.build_transform({"kind": "anscombe", "params": {"gain": 8}})
    assert isinstance(t, T.AnscombeTransform) and t.gain == 8.0
    t = T.build_transform({"kind": "linear", "params": {"mx": 500}})
    assert isinstance(t, T.LinearClipTransform) and t.mx == 500.0
    with pytest.raises(ValueError):
        T.build_transform({"kind": "nope"})
    t = T.build_transform({"kind": "asinh", "params": {"scale": 16}})
    assert t.cfg["kind"] == "asinh" and t.cfg["params"]["scale"] == 16


def test_norm_constants_match_reference():
    g = np.load(os.path.join(GOLD, "transforms.npz"))
    assert T.AsinhTransform(scale=32.0)._norm == float(g["asinh_s32/norm"])
    assert T.AsinhTransform(offset=35.0, scale=32.0)._norm == float(g["asinh_s32_o35/norm"])
    t = T.AnscombeTransform(gain=8.0, read_noise=5.0, offset=100.0)
    assert t._norm == float(g["anscombe_g8_rn5_o100/norm"])
    assert t._c_inv == 1.0 / 8.0
    assert T.AnscombeTransform(unbiased_inverse=False)._c_inv == 3.0 / 8.0


def test_with_offset_semantics():
    base = T.build_transform({"kind": "asinh", "params": {"scale": 32}})
    sh = T.with_offset(base, 120.0)
    assert isinstance(sh, T.OffsetTransform)
    assert sh.offset == 120.0 and sh.scale == 32.0 and sh.max_count == 65535.0
    assert sh.cfg["params"]["offset"] == 120.0 and sh.cfg["base"] == base.cfg
    again = T.with_offset(sh, 7.0)                       # re-wrapping replaces, never nests
    assert again.base_transform.cfg == base.cfg and again.offset == 7.0
    rebuilt = T.build_transform(sh.cfg)
    assert isinstance(rebuilt, T.OffsetTransform) and rebuilt.offset == 120.0
    lin = T.build_transform({"kind": "linear", "params": {"mn": 10.0, "mx": 1010.0, "clip": 8.0}})
    sl = T.with_offset(lin, 50.0)
    assert (sl.mn, sl.mx) == (60.0, 1060.0) and "offset" not in sl.cfg["params"]
    with pytest.raises(ValueError):
        T.with_offset(T.AsinhTransform(), 1.0)           # no cfg: not built via build_transform
    spec = sh.native_struct()
    assert spec.wrapped == 1 and spec.wrap_offset == 120.0 and spec.kind == 0


def test_estimate_offset_and_calibrate():
    g = np.load(os.path.join(GOLD, "transforms.npz"))
    rng = np.random.default_rng(7)
    sample = rng.integers(0, 400, size=5000).astype(np.uint16)
    sample[::13] = 0
    assert T.estimate_offset(sample, percentile=1.0) == float(g["estimate_offset/p1"])
    assert T.estimate_offset(sample, percentile=0.1) == float(g["estimate_offset/p0.1"])
    assert T.estimate_offset(sample, 5.0, ignore_zeros=False) == float(
        g["estimate_offset/p5_keepzeros"])
    s = np.arange(0, 101, dtype=np.float32)
    assert T.estimate_offset(s, percentile=0) == 1.0
    assert T.estimate_offset(s, percentile=0, ignore_zeros=False) == 0.0
    cfg = {"kind": "asinh", "calibrate": {"offset": True, "offset_percentile": 10.0}}
    out = T.calibrate_transform(cfg, np.arange(1, 1001, dtype=np.float32))
    assert abs(out["params"]["offset"] - float(np.percentile(np.arange(1, 1001), 10.0))) < 1e-4
    assert "params" not in cfg
    assert T.calibrate_transform({"kind": "anscombe", "params": {"gain": 2}},
                                 np.zeros(10))["params"] == {"gain": 2}


def test_base_class_raises():
    t = T.IntensityTransform()
    for fn in (t.forward, t.inverse, t.inverse_float):
        with pytest.raises(NotImplementedError):
            fn(np.zeros(1))


def test_patch_grid_helpers():
    g = np.load(os.path.join(GOLD, "tiling.npz"))
    for s in (64, 65, 100, 116, 117, 256, 1024):
        img = inference._ShapeOnly((1, 1, s, s, s))
        assert inference.count_patches(img, 64, 12) == int(g[f"count/{s}"])
        ax = [st[0] for st in inference.generate_patch_starts(
            inference._ShapeOnly((1, 1, s, 64, 64)), 64, 12)]
        assert ax == g[f"starts_axis/{s}"].tolist()
    p = inference.add_padding(np.ones((3, 4, 5)), 6)
    assert p.shape == (6, 6, 6) and p.sum() == 60 and p[3:].sum() == 0
    with pytest.raises(ValueError):
        inference.build_volume_transform(T.build_transform({"kind": "asinh"}))
    tf = inference.build_volume_transform(T.build_transform({"kind": "asinh"}), offset=37)
    assert isinstance(tf, T.OffsetTransform) and tf.offset == 37.0


def test_unet_matches_reference_fixture():
    """Same seed -> same parameters (names, shapes, values) and the same fp32 CPU output as the
    reference UNet (fixture generated by importing the reference)."""
    with open(os.path.join(GOLD, "unet_state.json")) as f:
        st = json.load(f)
    torch.manual_seed(0)
    model = unet3d.UNet()
    model.eval()
    sd = model.state_dict()
    assert set(sd) == set(st["abs_sums"])
    for k, v in sd.items():
        assert list(v.shape) == st["shapes"][k]
        assert abs(float(v.double().abs().sum()) - st["abs_sums"][k]) <= 1e-9 * max(
            1.0, st["abs_sums"][k])
    assert sum(p.numel() for p in model.parameters()) == st["n_params"] == 12946785
    x = torch.randn(1, 1, 32, 32, 32, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = model(x).numpy()
    np.testing.assert_allclose(y, np.load(os.path.join(GOLD, "unet.npz"))["y"], atol=1e-5)


def test_unet_validation_and_shapes():
    for bad in (0, 1.5, True, "2"):
        with pytest.raises(ValueError):
            unet3d.UNet(width_multiplier=bad)
    m = unet3d.N2V2UNet()
    assert m.config["model"] == "N2V2UNet"
    assert any(k.endswith("maxpool_conv.0.kernel") for k in m.state_dict())
    m.eval()
    x = torch.randn(1, 1, 33, 32, 35)
    with torch.no_grad():
        assert m(x).shape == x.shape
        assert unet3d.UNet(trilinear=False)(x[:, :, :32, :32, :32]).shape == (1, 1, 32, 32, 32)


def test_load_model_round_trip(tmp_path):
    """Checkpoint dict format of the reference Trainer (train.py:453-460) incl. the N2V2 branch
    that raises NameError in the reference (inference.py:290-291)."""
    for cls in (unet3d.UNet, unet3d.N2V2UNet):
        torch.manual_seed(3)
        model = cls()
        tcfg = {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}}
        path = tmp_path / f"{cls.__name__}.pth"
        torch.save({"model": model.state_dict(), "model_config": model.config,
                    "transform": tcfg}, path)
        loaded, tf = inference.load_model(str(path), device="cpu")
        assert type(loaded) is cls and not loaded.training
        assert tf.cfg["kind"] == "asinh" and tf.scale == 32.0
        for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
            assert torch.equal(a, b), k
    bare = tmp_path / "bare.pth"
    torch.save(unet3d.UNet().state_dict(), bare)
    loaded, tf = inference.load_model(str(bare), device="cpu")
    assert isinstance(tf, T.AsinhTransform)
