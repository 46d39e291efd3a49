"""Shared synthetic-data helpers for the tests (deterministic, numpy only)."""
import numpy as np


def synth_volume(shape, seed=0, sigma=24.0, pedestal=37.0, as_u16=False):
    """Pedestal + a few blurred bright 'neurites' + N(0, sigma) noise."""
    rng = np.random.default_rng(seed)
    nz, ny, nx = shape
    zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    clean = np.full(shape, pedestal, dtype=np.float64)
    for _ in range(max(2, int(np.prod(shape)) // 40000)):
        p0 = rng.uniform(0, 1, 3) * np.array(shape)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        amp = np.exp(rng.uniform(np.log(100), np.log(3000)))
        # distance of every voxel to the line p0 + t d
        v = np.stack([zz - p0[0], yy - p0[1], xx - p0[2]], axis=-1)
        t = v @ d
        dist2 = np.sum(v * v, axis=-1) - t * t
        clean += amp * np.exp(-dist2 / (2 * 1.5 ** 2))
    noisy = clean + rng.normal(0, sigma, shape)
    if as_u16:
        return np.rint(np.clip(noisy, 0, 65535)).astype(np.uint16), clean
    return noisy.astype(np.float32), clean


def psnr(a, b, peak):
    mse = np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(peak ** 2 / mse)


def metric_inputs(seed, shape=(40, 36, 44)):
    """(pred_u16, pred_f32, raw_u16, target_f32, fg) of one synthetic example."""
    rng = np.random.default_rng(seed)
    clean = np.full(shape, 120.0)
    clean[10:20, 8:30, 12:18] = 2500.0
    clean[25:27, :, 20:22] = 9000.0
    raw = np.rint(np.clip(clean + rng.normal(0, 24, shape), 0, 65535)).astype(np.uint16)
    raw[:2] = 0                                           # zero padding outside the imaged volume
    target = np.clip(clean + rng.normal(0, 3, shape), 0, 65535).astype(np.float32)
    pred_f32 = np.clip(clean + rng.normal(0, 5, shape), 0, 65535).astype(np.float32)
    pred_f32[30, 30, 30] = 7000.0                         # a hallucinated bright background voxel
    pred_u16 = np.rint(pred_f32).astype(np.uint16)
    fg = clean > 1000
    return pred_u16, pred_f32, raw, target, fg
