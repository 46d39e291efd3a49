"""GPU parity of the tiled-inference path (gather / trim + overlap-add / finalise kernels and the
drop-in ``predict``) against the numpy oracle and the reference-generated fixtures; the U-Net on
ROCm against the reference's fp32 CPU output."""
import os

import numpy as np
import pytest
import torch

from oracle import host_oracle as H
from test_oracle_golden import TILING_CASES, tiling_volume

from aind_exaspim_image_compression import _native  # noqa: E402
from aind_exaspim_image_compression import inference
from aind_exaspim_image_compression.machine_learning import transforms as T
from aind_exaspim_image_compression.machine_learning import unet3d

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TF_CFG = {"kind": "offset", "base": {"kind": "asinh", "params": {"offset": 0.0, "scale": 32.0}},
          "params": {"offset": 37.0}}


class Identity(torch.nn.Module):
    def forward(self, x):
        return x


class Affine(torch.nn.Module):
    def forward(self, x):
        return x * 0.5 + 0.125


@pytest.mark.parametrize("case", sorted(TILING_CASES))
@pytest.mark.parametrize("model", ["identity", "affine"])
def test_predict_matches_oracle_and_reference(case, model):
    shape, patch, overlap, trim, batch = TILING_CASES[case]
    vol = tiling_volume(shape)
    net = (Identity() if model == "identity" else Affine()).cuda()
    got = inference.predict(vol, net, T.build_transform(TF_CFG), batch_size=batch,
                            patch_size=patch, overlap=overlap, trim=trim, verbose=False)
    assert got.dtype == np.uint16 and got.shape == shape
    fn = (lambda b: b) if model == "identity" else (lambda b: b * np.float32(0.5) + np.float32(0.125))
    want = H.predict(vol, fn, H.TransformOracle(TF_CFG), batch_size=batch, patch=patch,
                     overlap=overlap, trim=trim)
    np.testing.assert_array_equal(got, want)                      # bit-exact vs the oracle
    ref = np.load(os.path.join(GOLD, "tiling.npz"))[f"predict/{case}/{model}"]
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and np.mean(d > 0) < 1e-3                # reference: SVML sinh, DESIGN 4.2


def test_short_tail_batch_runs_at_full_size_in_eval_mode():
    """Six patches in batches of four: in eval mode the tail of two goes through the model as a
    full batch (one input shape for MIOpen); the result is the oracle's either way."""
    shape = (130, 97, 64)
    vol = tiling_volume(shape)
    seen = []

    class Recording(Affine):
        def forward(self, x):
            seen.append(x.shape[0])
            return super().forward(x)

    want = H.predict(vol, lambda b: b * np.float32(0.5) + np.float32(0.125),
                     H.TransformOracle(TF_CFG), batch_size=4)
    for mode, sizes in (("eval", [4, 4]), ("train", [4, 2])):
        seen.clear()
        net = Recording().cuda()
        net = net.eval() if mode == "eval" else net.train()
        got = inference.predict(vol, net, T.build_transform(TF_CFG), batch_size=4, verbose=False)
        assert seen == sizes
        np.testing.assert_array_equal(got, want)


def test_tile_kernels_directly(ctx):
    rng = np.random.default_rng(1)
    shape = (40, 37, 45)
    vol = rng.normal(size=shape).astype(np.float32)
    starts = np.array([[0, 0, 0], [26, 26, 26], [13, 26, 0]], dtype=np.int32)
    patch, trim = 32, 3
    d_vol = ctx.to_device(vol)
    d_batch = ctx.alloc(3 * patch ** 3 * 4)
    ctx.tile_gather(d_vol, shape, starts, patch, d_batch)
    ctx.sync()
    got = d_batch.download((3, patch, patch, patch), np.float32)
    for b, (z, y, x) in enumerate(starts):
        want = np.zeros((patch,) * 3, np.float32)
        sub = vol[z:z + patch, y:y + patch, x:x + patch]
        want[:sub.shape[0], :sub.shape[1], :sub.shape[2]] = sub
        np.testing.assert_array_equal(got[b], want)
    acc = ctx.alloc(vol.nbytes).zero()
    wgt = ctx.alloc(vol.nbytes).zero()
    ctx.tile_accumulate(d_batch, starts, patch, trim, acc, wgt, shape)
    ctx.sync()
    a, w = acc.download(shape, np.float32), wgt.download(shape, np.float32)
    wa, ww = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
    core = patch - 2 * trim
    for b, st in enumerate(starts):
        s = [c + trim for c in st]
        e = [min(c + core, d) for c, d in zip(s, shape)]
        wa[s[0]:e[0], s[1]:e[1], s[2]:e[2]] += got[b, trim:trim + e[0] - s[0],
                                                    trim:trim + e[1] - s[1], trim:trim + e[2] - s[2]]
        ww[s[0]:e[0], s[1]:e[1], s[2]:e[2]] += 1
    np.testing.assert_array_equal(a, wa)
    np.testing.assert_array_equal(w, ww)


def test_unet_on_rocm_matches_reference_cpu_output():
    torch.manual_seed(0)
    model = unet3d.UNet().cuda().eval()
    x = torch.randn(1, 1, 32, 32, 32, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = model(x.cuda()).cpu().numpy()
    ref = np.load(os.path.join(GOLD, "unet.npz"))["y"]
    np.testing.assert_allclose(y, ref, atol=2e-3, rtol=1e-3)      # fp32 MIOpen vs fp32 CPU


def test_ndhwc_shadow_matches_the_reference_cpu_output_and_leaves_the_model_alone():
    """predict's default fast path (round 4): the forward passes run through an NDHWC copy of the model with
    MIOpen's implicit-GEMM solvers.  Same fp32 arithmetic: the reference's CPU output at the existing
    tolerance; the caller's module keeps its layout; fast=False calls the module as given and the two
    volumes agree to the criterion of test_predict_with_unet_and_predict_patch."""
    torch.manual_seed(0)
    model = unet3d.UNet().cuda().eval()
    shadow = inference._ndhwc_shadow(model)
    w = next(p for p in shadow.parameters() if p.dim() == 5)
    w0 = next(p for p in model.parameters() if p.dim() == 5)
    assert w.is_contiguous(memory_format=torch.channels_last_3d) and w0.is_contiguous()
    x = torch.randn(1, 1, 32, 32, 32, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = shadow(x.cuda()).cpu().numpy()
    np.testing.assert_allclose(y, np.load(os.path.join(GOLD, "unet.npz"))["y"], atol=2e-3, rtol=1e-3)
    tf = T.build_transform(TF_CFG)
    vol = tiling_volume((64, 116, 116), seed=4)
    a = inference.predict(vol, model, tf, batch_size=4, verbose=False).astype(np.int32)
    b = inference.predict(vol, model, tf, batch_size=4, verbose=False, fast=False).astype(np.int32)
    assert np.mean(np.abs(a - b) > 1) < 1e-3
    assert w0.is_contiguous() and not model.training


@pytest.mark.parametrize("shape", [(3, 32, 16, 16, 16), (2, 64, 8, 9, 10), (2, 512, 4, 4, 4), (1, 128, 5, 3, 2),
                                   (33, 32, 4, 4, 4), (2, 96, 8, 8, 8)])
def test_fused_groupnorm_leakyrelu_on_ndhwc(shape):
    """The BM4DNet stage's GroupNorm + LeakyReLU pairs as one NDHWC kernel pair (csrc/nn_kernels.hip) against
    the framework's two modules: fp64 statistics here, Welford in fp32 there -- agreement to a few ulp of
    the normalised values; the fused result is a deterministic function of its input; 96 channels (24 float4
    lanes do not divide a workgroup) take the framework's path through the same module."""
    b, c = shape[:2]
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 3 + 1.5).cuda().contiguous(memory_format=torch.channels_last_3d)
    norm = torch.nn.GroupNorm(8, c).cuda()
    with torch.no_grad():
        norm.weight.copy_(torch.randn(c, generator=g))
        norm.bias.copy_(torch.randn(c, generator=g))
    act = torch.nn.LeakyReLU(0.01)
    fused = inference.FusedGroupNormLeakyReLU(norm, act).eval()
    with torch.no_grad():
        want = act(norm(x.contiguous())).cpu().numpy()          # the framework's NCDHW path
        got1 = fused(x.clone()).cpu().numpy()
        got2 = fused(x.clone()).cpu().numpy()
    np.testing.assert_allclose(got1, want, atol=2e-5, rtol=2e-5)
    np.testing.assert_array_equal(got1, got2)
    with torch.no_grad():                                        # other layouts / modes: the framework's modules
        plain = fused(x.contiguous().clone())
    np.testing.assert_allclose(plain.cpu().numpy(), want, atol=1e-6, rtol=1e-6)
    # with the preceding convolution's bias folded in: lrelu(GN(x + bias)), no pass for the addition
    bias = torch.nn.Parameter((torch.randn(c, generator=g) * 2).cuda())
    with_bias = inference.FusedGroupNormLeakyReLU(norm, act, bias).eval()
    with torch.no_grad():
        want_b = act(norm(x.contiguous() + bias.view(1, -1, 1, 1, 1))).cpu().numpy()
        np.testing.assert_allclose(with_bias(x.clone()).cpu().numpy(), want_b, atol=3e-5, rtol=3e-5)
        np.testing.assert_allclose(with_bias(x.contiguous().clone()).cpu().numpy(), want_b, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("shape", [(2, 32, 16, 16, 16), (3, 64, 9, 7, 10), (1, 256, 2, 2, 2), (2, 8, 5, 6, 7)])
def test_ndhwc_maxpool_and_trilinear_upsample(shape):
    """MaxPool3d(2) and Upsample(2, trilinear, align_corners) on NDHWC tensors (csrc/nn_kernels.hip) against
    PyTorch's kernels on the NCDHW copy: the maximum exactly, the interpolation to rounding (same formula,
    same fp32 ratio; the framework's build may contract multiply-adds)."""
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape))).cuda()
    xl = x.contiguous(memory_format=torch.channels_last_3d)
    pool, up = torch.nn.MaxPool3d(2), torch.nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
    with torch.no_grad():
        for inner, exact in ((pool, True), (up, False)):
            mod = inference._ResampleNDHWC(inner).eval()
            assert mod.kind is not None
            got, want = mod(xl), inner(x)
            assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last_3d)
            if exact:
                assert torch.equal(got, want)
            else:
                np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), atol=2e-6, rtol=2e-6)
            np.testing.assert_allclose(mod(x).cpu().numpy(), want.cpu().numpy(), atol=2e-6, rtol=2e-6)   # NCDHW in: the framework's path
    assert inference._ResampleNDHWC(torch.nn.MaxPool3d(3)).kind is None
    assert inference._ResampleNDHWC(torch.nn.Upsample(scale_factor=2, mode="nearest")).kind is None


def test_shadow_fuses_every_norm_pair_and_resamples_on_ndhwc():
    torch.manual_seed(0)
    model = unet3d.UNet().cuda().eval()
    shadow = inference._ndhwc_shadow(model)
    fused = [m for m in shadow.modules() if isinstance(m, inference.FusedGroupNormLeakyReLU)]
    assert len(fused) == 18 and not any(isinstance(m, inference.FusedGroupNormLeakyReLU) for m in model.modules())
    assert not any(m.training for m in fused)            # the copies of an eval-mode model take the fused kernels
    assert sum(isinstance(m, inference._ResampleNDHWC) and m.kind is not None for m in shadow.modules()) == 8
    assert len(model.state_dict()) == len(shadow.state_dict())
    calls = []
    real = _native.Context.groupnorm_lrelu_ndhwc
    _native.Context.groupnorm_lrelu_ndhwc = lambda self, *a, **k: (calls.append(a[4]), real(self, *a, **k))[1]
    try:
        with torch.no_grad():
            shadow(torch.zeros(1, 1, 16, 16, 16, device="cuda"))
    finally:
        _native.Context.groupnorm_lrelu_ndhwc = real
    assert len(calls) == 18
    plain = inference._ndhwc_shadow(model, fuse=False)
    x = torch.randn(2, 1, 32, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        a, b = shadow(x).cpu().numpy(), plain(x).cpu().numpy()
    np.testing.assert_allclose(a, b, atol=2e-4, rtol=1e-4)


def test_n2v2_on_rocm_and_shape_contract():
    """N2V2UNet on ROCm against the reference's fp32 CPU output (fixture generated by importing
    the reference), and the shape contract of reference unet3d.py:574-590 at sizes 64 and 65."""
    torch.manual_seed(0)
    model = unet3d.N2V2UNet().cuda().eval()
    gold = np.load(os.path.join(GOLD, "n2v2.npz"))
    for name, shape in (("cube32", (1, 1, 32, 32, 32)), ("odd", (1, 1, 33, 32, 35))):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            y = model(x.cuda()).cpu().numpy()
        np.testing.assert_allclose(y, gold[name], atol=2e-3, rtol=1e-3)
    for net in (model, unet3d.UNet().cuda().eval()):
        for size in (64, 65):
            x = torch.randn(1, 1, size, size, size, device="cuda")
            with torch.no_grad():
                assert net(x).shape == x.shape


def test_predict_with_unet_and_predict_patch():
    torch.manual_seed(0)
    model = unet3d.UNet().cuda().eval()
    tf = T.build_transform(TF_CFG)
    vol = tiling_volume((70, 64, 64), seed=3)
    out = inference.predict(vol, model, tf, batch_size=4, verbose=False)
    assert out.shape == vol.shape and out.dtype == np.uint16
    assert np.all(out[:5] == 37)                                   # reference quirk kept
    p = inference.predict_patch(vol[:64], model, tf)
    assert p.shape == (64, 64, 64) and p.dtype == np.uint16
    # core of the first patch below the seam with the second patch (z < 52 + 5):
    # predict == predict_patch up to fp32 conv reductions
    a = out[5:57, 5:59, 5:59].astype(np.int32)
    b = p[5:57, 5:59, 5:59].astype(np.int32)
    assert np.mean(np.abs(a - b) > 1) < 1e-3


def test_quick_start_gives_the_default_models_volume():
    """inference.quick_start (NDHWC weights + MIOpen's FAST find mode for one-off volumes) changes the
    solver, not the arithmetic: predict() returns the default model's volume up to fp32 summation order
    (the seeded random-init U-Net and the sinh of the inverse transform amplify it: the criterion is the
    one of test_predict_with_unet_and_predict_patch).  The find mode itself only takes effect in a process
    that has not run a convolution yet (tools/dbg/quick_start_1024.py measures that: 30.7 s against
    43.5 s for one 1024^3 volume)."""
    tf = T.build_transform(TF_CFG)
    vol = tiling_volume((64, 116, 116), seed=4)
    outs = []
    for prep in (lambda m: m, inference.quick_start):
        torch.manual_seed(0)
        model = prep(unet3d.UNet().cuda().eval())
        outs.append(inference.predict(vol, model, tf, batch_size=4, verbose=False).astype(np.int32))
    d = np.abs(outs[0] - outs[1])
    assert np.mean(d > 1) < 1e-3, (int(d.max()), float(np.mean(d > 1)), float(np.mean(d > 0)))


def test_chunk_byte_histograms_and_cratio(ctx):
    """Row f-1 front end: per-chunk byte-plane histograms (integer-exact vs numpy), the entropy
    rate proxy, and compute_cratio's chunk walk with a stand-in codec."""
    from aind_exaspim_image_compression.utils import img_util
    rng = np.random.default_rng(4)
    vol = (37 + rng.normal(0, 24, (70, 64, 100))).clip(0, 65535).round().astype(np.uint16)
    vol[10:20, 5:9, 50:60] += 3000
    got = img_util.chunk_byte_histograms(vol, (64, 64, 64))
    want = []
    for z0 in range(0, 70, 64):
        for y0 in range(0, 64, 64):
            for x0 in range(0, 100, 64):
                c = vol[z0:z0 + 64, y0:y0 + 64, x0:x0 + 64].reshape(-1)
                want.append([np.bincount(c & 255, minlength=256), np.bincount(c >> 8, minlength=256)])
    np.testing.assert_array_equal(got, np.array(want, dtype=np.uint32))
    assert got.sum() == 2 * vol.size

    class Half:                                   # stand-in codec: "compresses" to half the bytes
        def encode(self, chunk):
            return bytes(chunk.nbytes // 2)
    assert img_util.compute_cratio(vol, Half()) == 2.0
    assert img_util.compute_cratio(vol[None, None], Half()) == 2.0
    smooth = np.full((64, 64, 64), 1000, np.uint16)
    assert img_util.shuffled_entropy_cratio(vol) < img_util.shuffled_entropy_cratio(smooth)
    assert 1.5 < img_util.shuffled_entropy_cratio(vol) < 4.0
