"""The reference's two evaluation loops on the device operators (SURVEY.md section 8 row a-J).

``evaluate_blocks`` is the per-block loop of the reference's supervised evaluator (reference
``evaluate.py:88-112``): volume transform with the block's own background offset, ``predict``,
then ``compute_cratio`` and ``ssim3D(noise, denoised, data_range=max(noise))``.

``compare_with_bm4d`` is the per-patch loop of its unsupervised evaluator (reference
``evaluate.py:172-217``): the network's ``predict_patch`` against
``np.maximum(bm4d(noise, 10), 0).astype(int)`` on the patch with ``trim`` voxels removed per
side, scored by compression ratio, SSIM, mean and maximum absolute error.  The reference calls
BM4D once per patch inside the loop; here all crops go through ONE batched device call, which
changes nothing per patch (volumes of a batch are independent: tests/test_bm4d_gpu.py).

``codec`` is any object with ``encode(ndarray) -> bytes`` (the reference passes numcodecs' Blosc,
``evaluate.py:40``); without one the byte-shuffled order-0 entropy proxy of
``utils.img_util.shuffled_entropy_cratio`` is reported under the same keys (a rate proxy, not a
Blosc byte count).
File handling, plotting and CSV output of the reference classes are out of scope.
"""
import numpy as np

from aind_exaspim_image_compression import bm4d as _bm4d
from aind_exaspim_image_compression import inference
from aind_exaspim_image_compression.machine_learning import transforms as _transforms
from aind_exaspim_image_compression.utils import img_util


def _cratio(img, codec):
    if codec is not None:
        return img_util.compute_cratio(img, codec)
    return img_util.shuffled_entropy_cratio(np.clip(img, 0, 65535).astype(np.uint16))


def evaluate_blocks(noise_imgs, model, transform, codec=None, raw_input=True, batch_size=32):
    """``{block_id: {"cratio", "ssim", "cratio_noise"}}`` for a dict of uint16 volumes.

    Each volume may carry the reference's leading singleton axes ``(1, 1, Z, Y, X)`` or
    ``(1, Z, Y, X)`` (``img_util.read(path)[0]`` there)."""
    rows = {}
    for block_id in sorted(noise_imgs):
        noise = np.asarray(noise_imgs[block_id])
        vol = noise.reshape(noise.shape[-3:])
        tf = inference.build_volume_transform(transform, vol) if raw_input else transform
        denoised = inference.predict(vol, model, tf, batch_size=batch_size, verbose=False)
        rows[block_id] = {
            "cratio": _cratio(denoised, codec),
            "cratio_noise": _cratio(vol, codec),
            "ssim": float(img_util.ssim3D(vol, denoised, data_range=np.max(vol))),
        }
    return rows


def compare_with_bm4d(patches, model, transform, codec=None, offset=None, sigma=10.0, trim=5,
                      keep_images=False):
    """Metrics of the network against BM4D for a batch ``patches[N, Z, Y, X]`` (uint16 counts).

    Returns the reference's dict of per-patch lists: ``cratio``, ``cratio_noise``, ``cratio_gt``,
    ``ssim_noise``, ``ssim_gt``, ``l1_gt``, ``lmax_gt``.  ``offset``: the brain's background
    offset for raw-input models (reference ``evaluate.py:188-193``); None keeps ``transform``.
    ``keep_images`` adds the lists ``denoised_gt`` / ``denoised`` (the images that were scored)."""
    patches = np.asarray(patches)
    if patches.ndim == 3:
        patches = patches[None]
    if patches.ndim != 4:
        raise ValueError("compare_with_bm4d expects patches[N, Z, Y, X]")
    tf = _transforms.with_offset(transform, float(offset)) if offset is not None else transform
    crop = (slice(None),) + (slice(trim, -trim),) * 3 if trim else (slice(None),) * 4
    noise = np.ascontiguousarray(patches[crop])
    gt = np.maximum(_bm4d.bm4d(noise, sigma), 0).astype(int)          # one device call for all N
    out = {k: [] for k in ("cratio_noise", "cratio_gt", "cratio", "ssim_noise", "ssim_gt", "l1_gt",
                           "lmax_gt")}
    for i in range(patches.shape[0]):
        denoised = inference.predict_patch(patches[i], model, tf)[crop[1:]]
        out["cratio"].append(_cratio(denoised, codec))
        out["cratio_noise"].append(_cratio(noise[i], codec))
        out["cratio_gt"].append(_cratio(gt[i], codec))
        out["ssim_noise"].append(float(img_util.ssim3D(noise[i], denoised)))
        out["ssim_gt"].append(float(img_util.ssim3D(gt[i], denoised)))
        out["l1_gt"].append(img_util.compute_mae(gt[i], denoised))
        out["lmax_gt"].append(img_util.compute_lmax(gt[i], denoised))
        if keep_images:
            out.setdefault("denoised_gt", []).append(gt[i])
            out.setdefault("denoised", []).append(denoised)
    return out
