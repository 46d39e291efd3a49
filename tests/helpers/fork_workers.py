"""Helper program of tests/test_fork_gpu.py (run as a fresh process, never imported by pytest):
the reference's process contract around bm4d -- scripts/precompute.py:215-222 forks a
ProcessPoolExecutor whose workers each call bm4d(raw, sigma) on one 64^3 patch
(machine_learning/data_handling.py:332; also data_handling.py:1325-1330).

    python fork_workers.py clean <out.npy>     parent never touches the GPU; 2 forked workers,
                                               3 patches -> teachers saved to out.npy
    python fork_workers.py dirty               parent initialises HIP first, then forks: the
                                               worker must raise NativeError (exit code 0 if it did)
    python fork_workers.py broker <out.npy> [workers=8] [patches=24]
                                               EXABM4D_BROKER=1: the forked workers never touch the GPU, one
                                               owner process coalesces their single-patch calls
    python fork_workers.py devices <out.npy> <d0,d1,..> [patches=12]
                                               denoise_patches(batch, devices=[...]) from a parent that never
                                               touches the GPU
"""
import multiprocessing
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "aind-exaspim-image-compression_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SIGMA = 24.0


def patch(i):
    from util import synth_volume
    return synth_volume((64, 64, 64), seed=100 + i)[0]


def teacher(i):
    from bm4d import bm4d                        # the reference's import line (data_handling.py:12)
    return os.getpid(), np.clip(bm4d(patch(i), SIGMA), 0, 65535.0)


def try_teacher(i):
    from aind_exaspim_image_compression._native import NativeError
    try:
        teacher(i)
    except NativeError as e:
        return "NativeError: " + str(e)
    return "no error"


def main():
    mode = sys.argv[1]
    fork = multiprocessing.get_context("fork")
    if mode == "clean":
        with ProcessPoolExecutor(max_workers=2, mp_context=fork) as pool:
            res = list(pool.map(teacher, range(3)))
        pids = {pid for pid, _ in res}
        assert os.getpid() not in pids and 1 <= len(pids) <= 2
        np.save(sys.argv[2], np.stack([t for _, t in res]))
        return 0
    if mode == "broker":
        workers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
        n = int(sys.argv[4]) if len(sys.argv) > 4 else 24
        os.environ["EXABM4D_BROKER"] = "1"
        os.environ["EXABM4D_BROKER_IDLE"] = "2"
        with ProcessPoolExecutor(max_workers=workers, mp_context=fork) as pool:
            res = list(pool.map(teacher, range(n)))
        assert os.getpid() not in {pid for pid, _ in res}
        np.save(sys.argv[2], np.stack([t for _, t in res]))
        from aind_exaspim_image_compression import _native
        assert _native._hip_owner_pid is None         # neither the parent ...
        return 0
    if mode == "devices":
        devices = [int(d) for d in sys.argv[3].split(",")]
        n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
        from aind_exaspim_image_compression import _native
        from aind_exaspim_image_compression.bm4d import denoise_patches
        batch = np.stack([patch(i) for i in range(n)])
        out = denoise_patches(batch, SIGMA, devices=devices)
        assert _native._hip_owner_pid is None         # the parent stayed off the GPU
        np.save(sys.argv[2], out)
        return 0
    if mode == "dirty":
        from aind_exaspim_image_compression import _native
        _native.context(0)                       # the parent initialises HIP ...
        with ProcessPoolExecutor(max_workers=1, mp_context=fork) as pool:
            msg = pool.submit(try_teacher, 0).result(timeout=120)      # ... then forks
        print(msg)
        return 0 if msg.startswith("NativeError") and "fork" in msg else 3
    return 2


if __name__ == "__main__":
    sys.exit(main())
