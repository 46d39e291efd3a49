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
        assert psnr(got[i], want, 1000.0) > 80.0


def test_fork_after_hip_init_fails_loudly():
    r = subprocess.run([sys.executable, HELPER, "dirty"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert "NativeError" in r.stdout and "fork" in r.stdout
