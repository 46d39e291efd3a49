"""BM4DNet: the 3-D U-Net "learned shrinkage" stage (PyTorch-ROCm / MIOpen only).

Drop-in for the reference ``machine_learning/unet3d.py``: same class names, constructor
arguments, ``config`` property, forward semantics and -- so that reference checkpoints load
unchanged -- the same ``state_dict`` keys and parameter creation order (a model built under the
same ``torch.manual_seed`` has bit-identical initial weights; pinned by
``tests/golden/unet_state.json``).  BASELINE.json's north star keeps this stage on PyTorch; no
hand-written kernels here.

Architecture (reference unet3d.py:20-134, :392-475): residual U-Net with channel widths
32/64/128/256/512 x width_multiplier, four 2x down-samplings (max-pool, or for N2V2 an
anti-aliased max-blur-pool), four 2x up-samplings (trilinear, align_corners=True, or a stride-2
transposed convolution), double 3^3 convolutions with GroupNorm(gcd(8, C)) and LeakyReLU(0.01),
1^3 output convolution.
"""
from math import gcd
from numbers import Real

import torch
import torch.nn as nn
import torch.nn.functional as F

_BASE_WIDTHS = (32, 64, 128, 256, 512)


def _conv_norm_act(cin, cout, kernel_size):
    return [
        nn.Conv3d(cin, cout, kernel_size=kernel_size, padding=1),
        nn.GroupNorm(gcd(8, cout), cout),
        nn.LeakyReLU(negative_slope=0.01, inplace=True),
    ]


class DoubleConv(nn.Module):
    """(Conv3d -> GroupNorm -> LeakyReLU) x 2 (reference unet3d.py:137-208)."""

    def __init__(self, in_channels, out_channels, mid_channels=None, kernel_size=3):
        super().__init__()
        mid = mid_channels or out_channels
        self.double_conv = nn.Sequential(
            *_conv_norm_act(in_channels, mid, kernel_size),
            *_conv_norm_act(mid, out_channels, kernel_size),
        )

    def forward(self, x):
        return self.double_conv(x)


class Down(nn.Module):
    """MaxPool3d(2) then DoubleConv (reference unet3d.py:211-255)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool3d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.maxpool_conv(x)


def _pad_to(x, ref_shape):
    """Zero-pad the three spatial axes of ``x`` up to ``ref_shape`` (split low/high like the
    reference: floor on the low side)."""
    dd, dh, dw = (int(r) - int(s) for r, s in zip(ref_shape, x.shape[2:]))
    if dd == 0 and dh == 0 and dw == 0:
        return x
    return F.pad(x, [dw // 2, dw - dw // 2, dh // 2, dh - dh // 2, dd // 2, dd - dd // 2])


def _make_upsampler(in_channels, trilinear):
    if trilinear:
        return nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
    return nn.ConvTranspose3d(in_channels, in_channels // 2, kernel_size=2, stride=2)


class Up(nn.Module):
    """Upsample, pad to the skip tensor, concatenate, DoubleConv (reference unet3d.py:258-342)."""

    def __init__(self, in_channels, out_channels, trilinear=True):
        super().__init__()
        self.up = _make_upsampler(in_channels, trilinear)
        if trilinear:
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        x1 = _pad_to(self.up(x1), x2.shape[2:])
        return self.conv(torch.cat([x2, x1], dim=1))


class OutConv(nn.Module):
    """1x1x1 output convolution (reference unet3d.py:345-389)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


class UNet(nn.Module):
    """Residual 3-D U-Net (reference unet3d.py:20-134)."""

    def __init__(self, width_multiplier=1, trilinear=True, residual=True):
        super().__init__()
        bad = (isinstance(width_multiplier, bool) or not isinstance(width_multiplier, Real)
               or width_multiplier < 1 or not float(width_multiplier).is_integer())
        if bad:
            raise ValueError("width_multiplier must be a positive integer")
        self.width_multiplier = int(width_multiplier)
        self.channels = [c * self.width_multiplier for c in _BASE_WIDTHS]
        self.trilinear = trilinear
        self.residual = residual
        c = self.channels
        f = 2 if trilinear else 1
        # creation order == the reference's, so seeded initialisation is identical
        self.inc = DoubleConv(1, c[0])
        self.down1 = Down(c[0], c[1])
        self.down2 = Down(c[1], c[2])
        self.down3 = Down(c[2], c[3])
        self.down4 = Down(c[3], c[4] // f)
        self.up1 = Up(c[4], c[3] // f, trilinear)
        self.up2 = Up(c[3], c[2] // f, trilinear)
        self.up3 = Up(c[2], c[1] // f, trilinear)
        self.up4 = Up(c[1], c[0], trilinear)
        self.outc = OutConv(c[0], 1)

    @property
    def config(self):
        """Constructor arguments, as stored in checkpoints (reference unet3d.py:93-100)."""
        return {"width_multiplier": self.width_multiplier, "trilinear": self.trilinear,
                "residual": self.residual}

    def _encode(self, x):
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        return x1, x2, x3, x4, self.down4(x4)

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        d = self.up1(x5, x4)
        d = self.up2(d, x3)
        d = self.up3(d, x2)
        d = self.up4(d, x1)
        logits = self.outc(d)
        return x + logits if self.residual else logits


class MaxBlurPool3D(nn.Module):
    """Anti-aliased pooling: max-pool stride 1, replicate pad, depthwise 3^3 binomial blur with
    stride 2 (reference unet3d.py:493-535).  The kernel is a persistent buffer named ``kernel``
    so that N2V2 checkpoints load."""

    def __init__(self, channels):
        super().__init__()
        k1 = torch.tensor([1.0, 2.0, 1.0])
        k3 = k1[:, None, None] * k1[None, :, None] * k1[None, None, :]
        k3 = k3 / k3.sum()
        self.register_buffer("kernel", k3[None, None].repeat(channels, 1, 1, 1, 1),
                             persistent=True)
        self.channels = channels
        self.pool = nn.MaxPool3d(2, stride=1)

    def forward(self, x):
        x = F.pad(self.pool(x), [1, 1, 1, 1, 1, 1], mode="replicate")
        return F.conv3d(x, self.kernel, stride=2, groups=self.channels)


class DownBlur(nn.Module):
    """MaxBlurPool3D then DoubleConv (reference unet3d.py:478-490)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(MaxBlurPool3D(in_channels),
                                          DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.maxpool_conv(x)


class UpNoSkip3D(nn.Module):
    """Upsampling block without a skip connection (reference unet3d.py:538-571)."""

    def __init__(self, in_channels, out_channels, trilinear=True):
        super().__init__()
        self.up = _make_upsampler(in_channels, trilinear)
        if trilinear:
            self.conv = DoubleConv(in_channels, out_channels, mid_channels=in_channels // 2)
        else:
            self.conv = DoubleConv(in_channels // 2, out_channels)

    def forward(self, x):
        return self.conv(self.up(x))


class N2V2UNet(UNet):
    """Noise2Void2 variant: blur-pooling instead of max-pooling and no top-resolution skip
    (reference unet3d.py:392-475)."""

    def __init__(self, width_multiplier=1, trilinear=True, residual=True):
        super().__init__(width_multiplier=width_multiplier, trilinear=trilinear,
                         residual=residual)
        c = self.channels
        f = 2 if trilinear else 1
        self.down1 = DownBlur(c[0], c[1])
        self.down2 = DownBlur(c[1], c[2])
        self.down3 = DownBlur(c[2], c[3])
        self.down4 = DownBlur(c[3], c[4] // f)
        self.up4 = UpNoSkip3D(c[1] // f, c[0], trilinear)

    def forward(self, x):
        _, x2, x3, x4, x5 = self._encode(x)
        d = self.up1(x5, x4)
        d = self.up2(d, x3)
        d = self.up3(d, x2)
        logits = _pad_to(self.outc(self.up4(d)), x.shape[2:])
        return x + logits if self.residual else logits

    @property
    def config(self):
        cfg = super().config
        cfg["model"] = "N2V2UNet"
        return cfg


def test_unets():
    """Shape contract of both nets (reference unet3d.py:574-590)."""
    for cls in (UNet, N2V2UNet):
        model = cls()
        model.eval()
        for size in (32, 33, 64, 65):
            x = torch.randn(1, 1, size, size, size)
            with torch.no_grad():
                y = model(x)
            assert y.shape == x.shape, (cls.__name__, size, tuple(y.shape))
