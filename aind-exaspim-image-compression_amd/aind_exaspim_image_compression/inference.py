"""GPU-resident tiled BM4DNet inference (drop-in for the reference ``inference.py``).

Same functions, signatures, defaults and return types as the reference
(``predict`` inference.py:28, ``predict_patch`` :119, ``load_model`` :255,
``build_volume_transform`` :302, helpers :178-252, :340-380).  What differs is where the work
happens: the reference transforms the volume with numpy, gathers patches with a thread pool,
copies every batch to and from the device and adds 8000 patches into host accumulators in a
Python loop (hot loops 1-2 of SURVEY.md section 3-A).  Here the volume is uploaded once; the
intensity transform, patch gather + zero padding, trim + overlap-add and the final
normalise -> inverse transform -> rint -> uint16 are HIP kernels of ``libexabm4d.so`` working on
buffers that stay in HBM, and only the uint16 result comes back.  The U-Net itself is PyTorch-ROCm.

Reference quirk kept on purpose (inference.py:91-103): the first ``trim``
voxels along every axis receive zero weight and come out as ``transform.inverse(0)``.
"""
import itertools
import os

import numpy as np
import torch

from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression.machine_learning.transforms import (
    build_transform,
    estimate_offset,
    with_offset,
)
from aind_exaspim_image_compression.machine_learning.unet3d import N2V2UNet, UNet


def _model_device(model):
    try:
        dev = next(model.parameters()).device
    except (StopIteration, AttributeError):
        dev = None
    if dev is None or dev.type != "cuda":
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return dev


_MIOPEN_DONE = False


def _miopen_defaults():
    """Once per process, before its first convolution (``predict`` / ``load_model`` call it): make MIOpen
    pick the tuned solvers for the BM4DNet U-Net without a search.
      * a per-user MIOpen user-db directory (``$XDG_CACHE_HOME/exabm4d/miopen/torch-<version>``) seeded with
        the find-db records shipped in ``miopen_db/`` (12 KB of text MIOpen wrote during one exhaustive
        search on an MI355X) -- unless the caller already chose ``MIOPEN_USER_DB_PATH``; searches the caller
        runs later (``tune_model``) persist there too, so they are paid once per machine, not per process;
      * ``MIOPEN_FIND_MODE=2`` (FAST) unless the caller set the variable: a find-db hit selects the recorded
        solver at once, a miss falls back to heuristics instead of timing every solver (17 s per process
        for this network in the default mode).
    ``EXABM4D_MIOPEN_DEFAULTS=0`` leaves MIOpen's environment alone.  Measured (tools/dbg/miopen_cache_probe.sh,
    U-Net forward 32 x 64^3 fp32): with the records first call 0.1 s, then 68 ms (51 TFLOP/s); without them
    and without this function 15 s, then 116 ms (30 TFLOP/s)."""
    global _MIOPEN_DONE
    if _MIOPEN_DONE:
        return
    _MIOPEN_DONE = True
    if os.environ.get("EXABM4D_MIOPEN_DEFAULTS", "1") == "0":
        return
    if "MIOPEN_USER_DB_PATH" not in os.environ:
        import shutil
        base = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"),
                            "exabm4d", "miopen", "torch-" + torch.__version__.replace("+", "_"))
        src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
        try:
            os.makedirs(os.path.join(base, "db"), exist_ok=True)
            os.makedirs(os.path.join(base, "kernels"), exist_ok=True)
            for name in os.listdir(src):
                dst = os.path.join(base, "db", name)
                if name.endswith(".txt") and not os.path.exists(dst):
                    shutil.copyfile(os.path.join(src, name), dst)
            os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(base, "db")
            os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(base, "kernels"))
        except OSError:
            pass                                   # read-only home: MIOpen's own defaults, FAST mode below
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")


class FusedGroupNormLeakyReLU(torch.nn.Module):
    """``GroupNorm`` followed by ``LeakyReLU`` as ONE module for inference on NDHWC tensors: the pair of the
    reference's ``DoubleConv`` (unet3d.py:137-208) through ``exabm4d_groupnorm_lrelu_ndhwc_dev`` -- statistics,
    normalisation and activation in two passes over the layout MIOpen's convolutions produce, in place on the
    convolution's output.  PyTorch's own GroupNorm wants NCDHW: per layer a layout copy in, statistics, apply,
    the activation and a layout copy back (45 % of the forward's kernel time).  Falls back to the framework's
    two modules for anything the kernels do not take (training, other dtypes / layouts / channel counts)."""

    def __init__(self, norm, act, conv_bias=None):
        """``conv_bias``: the bias of the convolution in front, taken over from it (the caller sets that
        convolution's ``bias`` to None): added inside the kernels instead of in a pass of its own."""
        super().__init__()
        self.norm, self.act = norm, act
        self.conv_bias = conv_bias
        self._ws = None

    def forward(self, x):
        n = self.norm
        fused = (not self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5
                 and x.is_contiguous(memory_format=torch.channels_last_3d) and not torch.is_grad_enabled()
                 and n.num_channels % 4 == 0 and (n.num_channels // n.num_groups) % 4 == 0
                 and 256 % (n.num_channels // 4) == 0 and n.num_groups <= 32 and x.shape[0] <= 65535)
        if not fused:
            if self.conv_bias is not None:
                x = x + self.conv_bias.view(1, -1, 1, 1, 1)
            return self.act(self.norm(x))
        b, c = int(x.shape[0]), int(x.shape[1])
        spatial = int(x.shape[2]) * int(x.shape[3]) * int(x.shape[4])
        need = int(_native.lib().exabm4d_groupnorm_workspace_bytes(b, spatial, c, n.num_groups))
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        ctx = _native.context(x.device.index or 0)
        ctx.groupnorm_lrelu_ndhwc(torch.cuda.current_stream(x.device).cuda_stream, x, x, b, spatial, c,
                                  n.num_groups, n.weight, n.bias, n.eps, self.act.negative_slope,
                                  self._ws, need, self.conv_bias)
        return x


def _all_equal(v, want):
    return all(x == want for x in (v if isinstance(v, (tuple, list)) else (v,)))


class _ResampleNDHWC(torch.nn.Module):
    """The U-Net's ``MaxPool3d(2)`` and ``Upsample(scale_factor=2, mode="trilinear", align_corners=True)`` on
    NDHWC tensors through ``libexabm4d`` (csrc/nn_kernels.hip: a float4 of channels per thread).  PyTorch's own
    kernels for the two walk an NDHWC tensor through generic strides (3.8 ms per call on this U-Net's tensors);
    anything else -- other parameters, layouts, dtypes, training -- runs ``inner`` on an NCDHW copy."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        m = inner
        if isinstance(m, torch.nn.MaxPool3d):
            ok = (_all_equal(m.kernel_size, 2) and _all_equal(m.stride if m.stride is not None else m.kernel_size, 2)
                  and _all_equal(m.padding, 0) and _all_equal(m.dilation, 1) and not m.ceil_mode
                  and not m.return_indices)
            self.kind = "pool" if ok else None
        elif isinstance(m, torch.nn.Upsample):
            ok = (m.mode == "trilinear" and m.align_corners is True and m.size is None
                  and m.scale_factor is not None and _all_equal(m.scale_factor, 2))
            self.kind = "up" if ok else None
        else:
            self.kind = None

    def forward(self, x):
        native = (self.kind is not None and not self.training and not torch.is_grad_enabled() and x.is_cuda
                  and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] % 4 == 0
                  and x.is_contiguous(memory_format=torch.channels_last_3d)
                  and (self.kind == "up" or min(x.shape[2:]) >= 2))
        if not native:
            return self.inner(x.contiguous())
        b, c, d, h, w = (int(v) for v in x.shape)
        out_dims = (d // 2, h // 2, w // 2) if self.kind == "pool" else (2 * d, 2 * h, 2 * w)
        y = torch.empty((b, c) + out_dims, dtype=torch.float32, device=x.device,
                        memory_format=torch.channels_last_3d)
        ctx = _native.context(x.device.index or 0)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if self.kind == "pool":
            ctx.maxpool2_ndhwc(stream, x, y, b, d, h, w, c)
        else:
            ctx.upsample2_trilinear_ndhwc(stream, x, y, b, d, h, w, c)
        return y


def _fuse_norm_act(module):
    """Replace every (GroupNorm, LeakyReLU) neighbour pair inside ``nn.Sequential`` containers of ``module`` by
    a ``FusedGroupNormLeakyReLU`` + ``Identity`` (same positions: the copy's parameters are the pair's), and
    put every ``MaxPool3d`` / ``Upsample`` behind ``_ResampleNDHWC``.  For the private copy ``_ndhwc_shadow`` makes;
    its ``state_dict`` keys are not the model's any more."""
    for name, child in list(module.named_children()):
        if isinstance(child, (torch.nn.MaxPool3d, torch.nn.Upsample)):
            setattr(module, name, _ResampleNDHWC(child).train(module.training))
        else:
            _fuse_norm_act(child)
    if isinstance(module, torch.nn.Sequential):
        for i in range(len(module) - 1):
            a, b = module[i], module[i + 1]
            if isinstance(a, torch.nn.GroupNorm) and isinstance(b, torch.nn.LeakyReLU) and a.affine:
                conv = module[i - 1] if i > 0 else None
                bias = None
                if isinstance(conv, torch.nn.Conv3d) and conv.bias is not None and conv.out_channels == a.num_channels:
                    bias, conv.bias = conv.bias, None             # added inside the fused kernels instead
                module[i] = FusedGroupNormLeakyReLU(a, b, bias).train(module.training)   # (a new module starts in training mode)
                module[i + 1] = torch.nn.Identity()
    return module


def _ndhwc_shadow(model, fuse=True):
    """An NDHWC (channels_last_3d) copy of an eval-mode fp32 module for the forward passes of one ``predict``
    call: MIOpen's implicit-GEMM solvers for NDHWC weights run this U-Net at 51 TFLOP/s against 30 for the
    default layout, and (``fuse``) its GroupNorm + LeakyReLU pairs run as the fused NDHWC kernels of
    ``libexabm4d`` instead of converting the layout there and back around PyTorch's GroupNorm.  The caller's
    model is not touched (52 MB copied per call); same fp32 arithmetic, results differ by summation order
    (tests at 2e-3 against the golden patch)."""
    import copy
    shadow = copy.deepcopy(model).to(memory_format=torch.channels_last_3d)
    return _fuse_norm_act(shadow) if fuse else shadow


def tune_model(model):
    """Opt-in MIOpen tuning for the BM4DNet stage on MI355X: NDHWC weights + exhaustive solver
    search (``torch.backends.cudnn.benchmark``, process-wide).  U-Net forward, 32 x 64^3 fp32:
    116 -> 69 ms (tools/dbg/unet_variants.py); same fp32 arithmetic, results differ by summation
    order only (4e-6).  Both halves are needed -- NDHWC weights WITHOUT the search fall on a slow
    default solver -- and the search runs once per input shape, so this is for long jobs; it is
    not applied by default and ``predict`` itself never changes the caller's model."""
    torch.backends.cudnn.benchmark = True
    if isinstance(model, torch.nn.Module):
        model.to(memory_format=torch.channels_last_3d)
    return model


def quick_start(model):
    """Opt-in for ONE-OFF volumes on a fresh process: NDHWC weights + MIOpen's FAST find mode
    (``MIOPEN_FIND_MODE=2``, set here unless the caller set the variable; it must happen before the
    process runs its first convolution).  MIOpen's default find mode spends 17 s on the first U-Net
    batch of a process (kernel selection by timing); FAST mode picks by heuristics in 0.2 s.  Measured on
    an MI355X, U-Net forward 32 x 64^3 fp32 (tools/dbg/miopen_modes.py): default mode, default layout:
    first call 16.8 s, then 117 ms; FAST + NDHWC: first call 0.2 s, then 122 ms; FAST with the default
    layout falls on a slow solver (316 ms) -- hence both halves here.  For one 1024^3 volume (250
    batches), measured on a fresh process: 30.7 s against 43.5 s (tools/dbg/quick_start_1024.py); for long jobs ``tune_model`` (39 s search, then 70 ms) wins.
    Same fp32 arithmetic in all three; results differ by summation order only."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    if isinstance(model, torch.nn.Module):
        model.to(memory_format=torch.channels_last_3d)
    return model


def predict(img, model, transform, batch_size=32, patch_size=64, overlap=12, trim=5,
            verbose=True, fast=True):
    """Denoise a 3-D image by overlapping-patch inference; returns uint16 counts.

    Parameters follow the reference (inference.py:28-67): ``img`` is a 3-D array (leading
    singleton axes are accepted), ``model`` a torch module (or any callable on a
    ``(B,1,P,P,P)`` float32 CUDA tensor), ``transform`` the IntensityTransform the model was
    trained with.  ``fast`` (not in the reference; default on): an eval-mode ``nn.Module`` runs its forward
    passes through an NDHWC copy of itself with MIOpen's tuned solvers (``_miopen_defaults``,
    ``_ndhwc_shadow``; fp32 throughout); ``fast=False`` calls ``model`` exactly as given."""
    _miopen_defaults()
    img = np.asarray(img)
    while img.ndim > 3:
        if img.shape[0] != 1:
            raise ValueError("predict expects a single 3-D volume")
        img = img[0]
    if img.ndim != 3:
        raise ValueError("predict expects a 3-D volume")
    shape = tuple(int(s) for s in img.shape)
    n = int(np.prod(shape))
    dev = _model_device(model)
    ctx = _native.context(dev.index or 0)
    run = model
    if fast and isinstance(model, torch.nn.Module) and not model.training and \
            all(p.dtype == torch.float32 for p in model.parameters()) and any(True for _ in model.parameters()):
        run = _ndhwc_shadow(model)

    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        ctx.set_stream(stream.cuda_stream)       # kernels and the model share one stream
        src_u16 = img.dtype == np.uint16
        host = np.ascontiguousarray(img if src_u16 else img.astype(np.float32))
        d_raw = torch.from_numpy(host.view(np.int16) if src_u16 else host).to(dev)
        vol = torch.empty(shape, dtype=torch.float32, device=dev)
        transform.forward_device(ctx, d_raw, vol, n, src_u16)          # inference.py:69
        del d_raw

        accum_pred = torch.zeros(shape, dtype=torch.float32, device=dev)   # inference.py:81-82
        accum_wgt = torch.zeros(shape, dtype=torch.float32, device=dev)
        starts = list(generate_patch_starts(_ShapeOnly((1, 1) + shape), patch_size, overlap))
        pbar = None
        if verbose:
            from tqdm import tqdm
            pbar = tqdm(total=len(starts), desc="Denoise")
        batch = torch.empty((batch_size, 1, patch_size, patch_size, patch_size),
                            dtype=torch.float32, device=dev)
        for b0 in range(0, len(starts), batch_size):
            chunk = np.asarray(starts[b0:b0 + batch_size], dtype=np.int32)
            nb = len(chunk)
            ctx.tile_gather(vol, shape, chunk, patch_size, batch)      # inference.py:153-168
            # A short last batch is run at full size (rows beyond nb hold the previous batch's
            # patches and are dropped): MIOpen then sees ONE input shape per volume instead of
            # paying a solver search -- seconds -- for the tail's.  Only for modules in eval mode,
            # whose output rows do not depend on the rest of the batch.
            full = nb < batch_size and b0 > 0 and not getattr(model, "training", True)
            with torch.no_grad():
                out = run(batch if full else batch[:nb])[:nb]          # inference.py:171-173
            out = out.to(torch.float32).contiguous()
            ctx.tile_accumulate(out, chunk, patch_size, trim, accum_pred, accum_wgt, shape)
            if pbar is not None:
                pbar.update(nb)
        del vol
        result = torch.empty(shape, dtype=torch.int16, device=dev)
        ctx.tile_finalize(transform.native_struct(), accum_pred, accum_wgt, result, n)
        stream.synchronize()
        out = result.cpu().numpy().view(np.uint16)
        ctx.reset_stream()
    del run
    if pbar is not None:
        pbar.close()
    return out


def predict_patch(patch, model, transform):
    """Denoise one patch (reference inference.py:119-150); uint16, same shape as the input."""
    patch = np.asarray(patch)
    shape = patch.shape[-3:]
    x = transform.forward(patch.reshape(shape))
    dev = _model_device(model)
    with torch.no_grad():
        pred = model(to_tensor(x, device=dev))
    return transform.inverse(pred[0, 0].float().cpu().numpy())


# --- helpers (reference inference.py:178-252) -------------------------------------------------
class _ShapeOnly:
    def __init__(self, shape):
        self.shape = shape


def add_padding(patch, patch_size):
    """Zero-pad a 3-D patch at the high end of each axis up to ``patch_size`` (host helper kept
    for API parity; ``predict`` pads inside the gather kernel)."""
    patch = np.asarray(patch)
    pad = [(0, patch_size - s) for s in patch.shape]
    return np.pad(patch, pad, mode="constant", constant_values=0)


def generate_patch_starts(img, patch_size, overlap):
    """Patch corner coordinates: ``range(0, dim - patch + stride, stride)`` per spatial axis of
    a ``(1, 1, D, H, W)`` image, z outermost."""
    stride = patch_size - overlap
    axes = [range(0, img.shape[a] - patch_size + stride, stride) for a in (2, 3, 4)]
    return itertools.product(*axes)


def count_patches(img, patch_size, overlap):
    stride = patch_size - overlap
    n = 1
    for a in (2, 3, 4):
        n *= len(range(0, img.shape[a] - patch_size + stride, stride))
    return n


def load_model(path, device="cuda"):
    """Load a checkpoint -> ``(model.eval(), transform)`` (reference inference.py:255-299).

    Accepts the current format ``{"model", "model_config", "transform"}`` and a bare legacy
    ``state_dict`` (transform then defaults to asinh).  Unlike the reference, ``N2V2UNet``
    checkpoints work (the reference forgets to import the class: inference.py:290-291)."""
    _miopen_defaults()
    ckpt = torch.load(path, map_location=device)
    if isinstance(ckpt, dict) and "model" in ckpt:
        state_dict = ckpt["model"]
        transform_cfg = ckpt.get("transform") or {"kind": "asinh"}
        model_cfg = dict(ckpt.get("model_config") or {})
    else:
        state_dict, transform_cfg, model_cfg = ckpt, {"kind": "asinh"}, {}
    cls = N2V2UNet if model_cfg.pop("model", "UNet") == "N2V2UNet" else UNet
    model = cls(**model_cfg)
    model.load_state_dict(state_dict)
    model.to(device)
    model.eval()
    return model, build_transform(transform_cfg)


def build_volume_transform(base_transform, img=None, percentile=0.1, offset=None):
    """Inference transform carrying a per-volume background offset (reference
    inference.py:302-337): a supplied ``offset`` is used as is, otherwise it is estimated from
    ``img``; ``ValueError`` if neither is given."""
    if offset is None:
        if img is None:
            raise ValueError("img is required when offset is not supplied")
        offset = estimate_offset(img, percentile=percentile, ignore_zeros=True)
    return with_offset(base_transform, offset)


def to_tensor(arr, device="cuda"):
    """numpy -> float tensor of shape (1, 1, D, H, W) on ``device`` (inference.py:340-359)."""
    arr = np.asarray(arr)
    while arr.ndim < 5:
        arr = arr[np.newaxis, ...]
    return torch.tensor(arr).to(device, dtype=torch.float)


def batch_to_tensor(arr, device="cuda"):
    """(B, D, H, W) numpy -> (B, 1, D, H, W) float tensor (inference.py:362-380)."""
    return to_tensor(np.asarray(arr)[:, np.newaxis, ...], device=device)
