"""Helper program of tests/test_fork_gpu.py (run as a fresh process, never imported by pytest):
the reference's process contract around bm4d -- scripts/precompute.py:215-222 forks a
ProcessPoolExecutor whose workers each call bm4d(raw, sigma) on one 64^3 patch
(machine_learning/data_handling.py:332; also data_handling.py:1325-1330).

    python fork_workers.py clean <out.npy>     parent never touches the GPU; 2 forked workers,
                                               3 patches -> teachers saved to out.npy
    python fork_workers.py dirty               parent initialises HIP first, then forks: the
                                               worker must raise NativeError (exit code 0 if it did)
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
