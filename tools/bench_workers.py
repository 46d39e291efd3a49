"""The reference's BM4D-at-scale pattern on one MI355X, three ways:

    python tools/bench_workers.py [patches=1000] [workers=16] [direct_workers=5]

  direct   `direct_workers` forked workers, each with its own HIP context, one bm4d(patch, sigma) per call
           (the drop-in as it is; a GPU box admits at most 6 processes on the card, hence 5)
  broker   `workers` forked workers that never touch the GPU + one owner process (EXABM4D_BROKER=1)
  batched  one denoise_patches(batch) call from one process
on the same `patches` 64^3 fp32 patches (the work shape of scripts/precompute.py:215-228); checks that all
three give the same teachers and prints one JSON line.  The parent never touches the GPU: every leg is a
child process."""
import json
import multiprocessing
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "aind-exaspim-image-compression_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SIGMA = 24.0
_BASE = None


def base():
    """32 distinct patches, cycled (with a per-index shift) to the requested count"""
    global _BASE
    if _BASE is None:
        from util import synth_volume
        _BASE = np.stack([synth_volume((64, 64, 64), seed=300 + i)[0] for i in range(32)])
    return _BASE


def patch(i):
    return base()[i % 32] + np.float32(i // 32)


def one(i):
    from bm4d import bm4d
    return np.clip(bm4d(patch(i), SIGMA), 0, 65535.0)


def checksum(i):
    return float(one(i).astype(np.float64).sum())


def pool_leg(n, workers, env):
    os.environ.update(env)
    fork = multiprocessing.get_context("fork")
    with ProcessPoolExecutor(max_workers=workers, mp_context=fork) as pool:
        list(pool.map(checksum, range(workers)))               # contexts / broker up, kernels loaded
        t0 = time.perf_counter()
        sums = list(pool.map(checksum, range(n), chunksize=1))
        dt = time.perf_counter() - t0
    if env.get("EXABM4D_BROKER") == "1":
        from aind_exaspim_image_compression import broker
        print("broker stats:", broker.stats(0), file=sys.stderr, flush=True)
    return dt, sums


def batched_leg(n):
    from aind_exaspim_image_compression.bm4d import denoise_patches
    batch = np.stack([patch(i) for i in range(n)])
    denoise_patches(batch[:32], SIGMA)
    t0 = time.perf_counter()
    out = denoise_patches(batch, SIGMA)
    dt = time.perf_counter() - t0
    return dt, [float(o.astype(np.float64).sum()) for o in out]


def leg(q, name, n, workers):
    try:
        if name == "batched":
            q.put((name, batched_leg(n)))
        elif name == "broker":
            q.put((name, pool_leg(n, workers, {"EXABM4D_BROKER": "1", "EXABM4D_BROKER_IDLE": "3"})))
        else:
            q.put((name, pool_leg(n, workers, {})))
    except Exception as e:
        q.put((name, repr(e)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    direct_workers = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    base()
    spawn = multiprocessing.get_context("spawn")
    res = {}
    for name, w in (("batched", 1), ("broker", workers), ("direct", direct_workers)):
        q = spawn.Queue()
        p = spawn.Process(target=leg, args=(q, name, n, w))
        p.start()
        got = q.get(timeout=1500)
        p.join(timeout=60)
        if isinstance(got[1], str):
            res[name] = {"error": got[1]}
            continue
        dt, sums = got[1]
        res[name] = {"seconds": round(dt, 4), "patches_per_s": round(n / dt, 1), "workers": w, "sums": sums}
    ref = res.get("batched", {}).get("sums")
    for name in res:
        if "sums" in res[name]:
            res[name]["equals_batched"] = bool(ref is not None and res[name]["sums"] == ref)
            del res[name]["sums"]
    if "seconds" in res.get("broker", {}) and "seconds" in res.get("batched", {}):
        res["broker_over_batched"] = round(res["broker"]["seconds"] / res["batched"]["seconds"], 2)
    res["patches"] = n
    print(json.dumps(res))


if __name__ == "__main__":
    main()
