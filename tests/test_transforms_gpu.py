"""GPU parity of the fused intensity-transform kernels (through the C-ABI / the drop-in
transform objects) against the numpy oracle and the reference-generated fixtures; plus the
reference's own tests/test_transforms.py cases re-run against the drop-in package."""
import os

import numpy as np
import pytest

from oracle import host_oracle as H
from test_oracle_golden import GRID, TRANSFORM_CFGS, U16, ulp_diff

from aind_exaspim_image_compression.machine_learning import transforms as T

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", sorted(TRANSFORM_CFGS))
def test_kernels_equal_oracle_bit_for_bit(name):
    """All 65536 uint16 inputs and a dense fp32 grid; float and uint16 outputs bit-exact."""
    cfg = TRANSFORM_CFGS[name]
    t, o = T.build_transform(cfg), H.TransformOracle(cfg)
    allu16 = np.arange(65536, dtype=np.uint16)
    np.testing.assert_array_equal(t.forward(allu16), o.forward(allu16))
    f32_in = np.linspace(-50.0, 70000.0, 200001, dtype=np.float32)
    np.testing.assert_array_equal(t.forward(f32_in), o.forward(f32_in))
    grid = np.linspace(-0.2, 1.2, 400001, dtype=np.float32)
    np.testing.assert_array_equal(t.inverse(grid), o.inverse(grid))          # integer indices
    np.testing.assert_array_equal(t.inverse_float(grid), o.inverse_float(grid))


@pytest.mark.parametrize("name", sorted(TRANSFORM_CFGS))
def test_kernels_against_reference_fixture(name):
    g = np.load(os.path.join(GOLD, "transforms.npz"))
    t = T.build_transform(TRANSFORM_CFGS[name])
    fwd, inv = t.forward(U16), t.inverse(GRID)
    if "asinh" in name:      # numpy's SVML fp32 arcsinh/sinh vs correctly rounded: DESIGN.md 4.2
        assert ulp_diff(fwd, g[f"{name}/forward_u16"]).max() <= 2
        q = np.abs(inv.astype(np.int32) - g[f"{name}/inverse_u16"].astype(np.int32))
        assert q.max() <= 1 and np.count_nonzero(q) <= 4
    else:
        np.testing.assert_array_equal(fwd, g[f"{name}/forward_u16"])
        np.testing.assert_array_equal(inv, g[f"{name}/inverse_u16"])


# ---- the reference's tests/test_transforms.py, against the drop-in objects --------------------
def test_ref_asinh_cases():
    t = T.AsinhTransform(offset=35, scale=32)
    vals = np.array([0, 100, 1000, 10000, 60000, 65535], dtype=np.float32)
    rec = t.inverse(t.forward(vals)).astype(np.float64)
    np.testing.assert_allclose(rec, vals, rtol=1e-2, atol=3)
    rec = t.inverse(t.forward(np.array([2500, 10000, 60000], np.float32))).astype(np.float64)
    assert np.all(np.diff(rec) > 1000)
    ys = t.forward(np.linspace(0, 65535, 500).astype(np.float32))
    assert np.all(np.diff(ys) > 0)
    assert abs(float(t.forward(np.array(65535.0))) - 1.0) < 1e-4
    assert abs(float(t.forward(np.array(35.0)))) < 0.05
    assert float(t.forward(np.array(0.0))) < 0.0


def test_ref_anscombe_cases():
    t = T.AnscombeTransform(gain=8, read_noise=5, offset=100, unbiased_inverse=False)
    vals = np.array([100, 500, 2000, 20000, 65535], dtype=np.float32)
    np.testing.assert_allclose(t.inverse(t.forward(vals)).astype(np.float64), vals, rtol=5e-3,
                               atol=3)
    ta, tu = T.AnscombeTransform(gain=8, unbiased_inverse=False), T.AnscombeTransform(gain=8)
    y = ta.forward(np.array([1000, 20000, 60000], dtype=np.float32))
    np.testing.assert_allclose(tu.inverse(y).astype(np.float64) - ta.inverse(y), 2.0, atol=1.0)
    t = T.AnscombeTransform(gain=8, read_noise=5, offset=0)
    assert np.all(np.diff(t.forward(np.linspace(0, 65535, 500).astype(np.float32))) > 0)
    t = T.AnscombeTransform(gain=8, read_noise=5, offset=100)
    assert abs(float(t.forward(np.array(65535.0))) - 1.0) < 1e-4
    t = T.AnscombeTransform(gain=1, read_noise=0, offset=0)
    x = np.array([0, 1, 10, 100, 1000], dtype=np.float32)
    np.testing.assert_allclose(t._gat(x), 2.0 * np.sqrt(x + 3.0 / 8.0), rtol=1e-5)


def test_ref_linear_and_offset_cases():
    t = T.LinearClipTransform(mn=35, mx=1000, clip=8)
    vals = np.array([35, 200, 1000, 5000], dtype=np.float32)
    np.testing.assert_allclose(t.inverse(t.forward(vals)).astype(np.float64), vals, rtol=1e-3,
                               atol=1)
    t = T.LinearClipTransform(mn=0, mx=1000, clip=8)
    rec = t.inverse(t.forward(np.array([9000, 30000, 60000], np.float32))).astype(np.float64)
    assert np.all(rec == rec[0]) and rec[0] < 9000
    base = T.build_transform({"kind": "asinh", "params": {"scale": 32}})
    sh = T.with_offset(base, 120.0)
    values = np.array([120.0, 152.0, 1120.0, 60120.0])
    np.testing.assert_array_equal(sh.forward(values), base.forward(values - 120.0))
    np.testing.assert_allclose(sh.inverse(sh.forward(values)), values, atol=1)
    np.testing.assert_array_equal(T.build_transform(sh.cfg).forward(values), sh.forward(values))
    ab = T.build_transform({"kind": "anscombe", "params": {"gain": 8, "read_noise": 5}})
    sa = T.with_offset(ab, 120.0)
    v2 = np.array([120.0, 500.0, 2000.0, 20000.0])
    np.testing.assert_array_equal(sa.forward(v2), ab.forward(v2 - 120.0))
    lin = T.build_transform({"kind": "linear", "params": {"mn": 10.0, "mx": 1010.0, "clip": 8.0}})
    sl = T.with_offset(lin, 50.0)
    v3 = np.array([60.0, 560.0, 1060.0], dtype=np.float32)
    np.testing.assert_allclose(sl.forward(v3), lin.forward(v3 - 50.0))


def test_shapes_and_empty():
    t = T.build_transform({"kind": "asinh"})
    x = np.arange(24, dtype=np.uint16).reshape(2, 3, 4)
    assert t.forward(x).shape == (2, 3, 4) and t.forward(x).dtype == np.float32
    assert t.inverse(t.forward(x)).dtype == np.uint16
    assert t.forward(np.zeros((0,), np.uint16)).shape == (0,)
    np.testing.assert_array_equal(t.inverse(t.forward(x)), x)


def test_large_u16_volume_takes_the_table_path_bit_exact():
    """>= 2^20 uint16 voxels with an asinh transform go through the 65536-entry device table
    (filled by the same device function): identical bits to direct evaluation and to the oracle."""
    cfg = TRANSFORM_CFGS["offset37_asinh_s32"]
    t, o = T.build_transform(cfg), H.TransformOracle(cfg)
    rng = np.random.default_rng(0)
    big = rng.integers(0, 65536, size=(1 << 20) + 13).astype(np.uint16)
    got = t.forward(big)
    np.testing.assert_array_equal(got, o.forward(big))
    np.testing.assert_array_equal(got[:4096], t.forward(big[:4096]))      # direct path
