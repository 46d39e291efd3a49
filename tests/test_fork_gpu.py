"""The boundary's process contract: the reference calls bm4d inside forked
ProcessPoolExecutor workers, one call per 64^3 patch (scripts/precompute.py:215-222,
machine_learning/data_handling.py:332, :1325-1330).  (a) forked workers of a parent that never
touched the GPU each create their own context and return teachers equal to the oracle's;
(b) workers forked AFTER the parent initialised HIP fail loudly (NativeError), they do not hang.
Both scenarios run in a fresh child process (tests/helpers/fork_workers.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from util import psnr, synth_volume

pytestmark = pytest.mark.gpu
HELPER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "fork_workers.py")


def test_forked_workers_each_own_a_context(oracle, tmp_path):
    out = tmp_path / "teachers.npy"
    r = subprocess.run([sys.executable, HELPER, "clean", str(out)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    assert got.shape == (3, 64, 64, 64) and got.dtype == np.float32
    for i in range(3):
        raw = synth_volume((64, 64, 64), seed=100 + i)[0]
        want = np.clip(oracle.bm4d(raw, 24.0), 0, 65535.0)
        np.testing.assert_array_equal(got[i], want)


def test_broker_coalesces_the_workers_single_patch_calls(oracle, tmp_path, monkeypatch):
    """Round 4: EXABM4D_BROKER=1 and nothing else changed -- eight forked workers (none of them ever opens a
    HIP context), 24 single-patch bm4d() calls, one GPU-owner process that batches what is pending.  Every
    volume of a batched call has its own fixed-point unit, so the teachers are the oracle's bit for bit,
    whatever a patch happened to be batched with."""
    monkeypatch.setenv("EXABM4D_BROKER_DIR", str(tmp_path))
    out = tmp_path / "teachers.npy"
    r = subprocess.run([sys.executable, HELPER, "broker", str(out), "8", "24"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stderr[-2000:], open(next(tmp_path.glob("*.log"))).read()[-2000:])
    got = np.load(out)
    assert got.shape == (24, 64, 64, 64)
    for i in (0, 7, 23):
        raw = synth_volume((64, 64, 64), seed=100 + i)[0]
        np.testing.assert_array_equal(got[i], np.clip(oracle.bm4d(raw, 24.0), 0, 65535.0))
    # the rest against the direct batched call of this process
    from aind_exaspim_image_compression.bm4d import denoise_patches
    batch = np.stack([synth_volume((64, 64, 64), seed=100 + i)[0] for i in range(24)])
    np.testing.assert_array_equal(got, denoise_patches(batch, 24.0))


def test_denoise_patches_over_several_devices(tmp_path):
    """denoise_patches(raw, sigma, devices=[...]): consecutive shares of the batch in fresh child processes,
    one per entry (two entries for the one GPU of this box: same mechanics as two GPUs), from a parent that
    stays off the GPU; equal to the single-device call."""
    out = tmp_path / "teachers.npy"
    r = subprocess.run([sys.executable, HELPER, "devices", str(out), "0,0", "5"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    from aind_exaspim_image_compression.bm4d import denoise_patches
    batch = np.stack([synth_volume((64, 64, 64), seed=100 + i)[0] for i in range(5)])
    np.testing.assert_array_equal(np.load(out), denoise_patches(batch, 24.0))


def test_fork_after_hip_init_fails_loudly():
    r = subprocess.run([sys.executable, HELPER, "dirty"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert "NativeError" in r.stdout and "fork" in r.stdout
