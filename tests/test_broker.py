"""Host logic of the multi-device drop-in for the reference's worker pattern (forked pool, one bm4d() call
per patch: scripts/precompute.py:215-228): device choice of a worker, batch splitting, the broker's batching
policy and its protocol -- the latter with the GPU call replaced by a stand-in INSIDE THIS TEST (the broker's
serve() runs in a thread with _native.context monkeypatched; the product has no such switch)."""
import multiprocessing
import os
import threading
import time

import numpy as np
import pytest

from aind_exaspim_image_compression import _native, bm4d as B, broker


def test_default_device_rules(monkeypatch):
    for var in ("EXABM4D_DEVICE", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    assert _native.worker_index() is None and _native.default_device(count=8) == 0      # the main process
    monkeypatch.setattr(multiprocessing.current_process(), "_identity", (11,))
    assert _native.worker_index() == 10
    assert _native.default_device(count=8) == 2 and _native.default_device(count=1) == 0
    assert _native.default_device(count=0) == 0                                          # unknown count
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert _native.default_device(count=8) == 5
    monkeypatch.setenv("EXABM4D_DEVICE", "3")
    assert _native.default_device(count=8) == 3
    monkeypatch.delenv("EXABM4D_DEVICE")
    monkeypatch.delenv("LOCAL_RANK")
    # sixteen workers of a pool on an eight-GPU node: two per device
    devs = []
    for w in range(1, 17):
        monkeypatch.setattr(multiprocessing.current_process(), "_identity", (w,))
        devs.append(_native.default_device(count=8))
    assert sorted(devs) == sorted(list(range(8)) * 2)


def test_visible_device_count_without_hip(monkeypatch):
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert _native.device_count_no_init() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert _native.device_count_no_init() == 0


def test_split_and_batch_plans():
    assert B.split_batch(1000, 8) == [(125 * i, 125 * (i + 1)) for i in range(8)]
    assert B.split_batch(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert B.split_batch(2, 8) == [(0, 1), (1, 2)] and B.split_batch(0, 4) == []
    # one device call per key in order of first arrival, arrival order inside, split at the voxel cap
    calls = broker.plan_batches([(0, "a", 5), (1, "b", 5), (2, "a", 5), (3, "a", 5), (4, "b", 11)], max_voxels=10)
    assert calls == [[0, 2], [3], [1], [4]]
    assert broker.plan_batches([]) == []


class _FakeContext:
    """Stand-in for the device: 'denoising' = x / 2 + sigma, per volume, in place at the addresses the broker
    hands over (the interface of Context.denoise_f32_host_v); records the batch sizes."""

    def __init__(self):
        self.batches = []

    def denoise_f32_host_v(self, in_addrs, out_addrs, shape, sigma, params=None, stages=2, clip=None):
        import ctypes
        self.batches.append(len(in_addrs))
        time.sleep(0.02)                                   # long enough for the other workers to queue up
        n = int(np.prod(shape))
        for a, b in zip(in_addrs, out_addrs):
            x = np.ctypeslib.as_array((ctypes.c_float * n).from_address(a))
            y = np.ctypeslib.as_array((ctypes.c_float * n).from_address(b))
            out = x * np.float32(0.5) + np.float32(sigma)
            y[...] = np.clip(out, clip[0], clip[1]) if clip is not None else out


def _worker(args):
    seed, n = args
    broker.enable()
    rng = np.random.default_rng(seed)
    ok = True
    for _ in range(n):
        x = rng.normal(100, 20, (16, 16, 16)).astype(np.float32)
        y = B.bm4d(x, 8.0)                                  # the reference's call, unchanged
        ok = ok and np.array_equal(y, x * np.float32(0.5) + np.float32(8.0))
    z = B.denoise_patches(np.stack([x, x + 1]), 8.0)       # a batch from one worker, clipped
    ok = ok and np.array_equal(z[1], np.clip((x + 1) * np.float32(0.5) + np.float32(8.0), 0, 65535))
    return ok


def test_broker_protocol_coalesces_forked_workers(monkeypatch, tmp_path):
    monkeypatch.setenv(broker.ENV_DIR, str(tmp_path))
    monkeypatch.setenv("EXABM4D_DEVICE", "0")
    fake = _FakeContext()
    monkeypatch.setattr(_native, "context", lambda device=None: fake)
    monkeypatch.setattr(_native, "new_context", lambda device=None: fake)
    broker._authkey(0, create=True)
    t = threading.Thread(target=broker.serve, kwargs=dict(device=0, idle=1.5, linger=0.01), daemon=True)
    t.start()
    for _ in range(200):
        if os.path.exists(broker.socket_path(0)):
            break
        time.sleep(0.01)
    with multiprocessing.get_context("fork").Pool(6) as pool:
        assert all(pool.map(_worker, [(s, 5) for s in range(6)], chunksize=1))
    t.join(timeout=10)
    assert not t.is_alive()                                 # left by itself once idle
    assert sum(fake.batches) == 6 * 5 + 6 * 2               # every patch once
    assert max(fake.batches) >= 3                           # ... and several workers' patches in one call
    assert not os.path.exists(broker.socket_path(0))


def test_batch_cap_cuts_the_clients_into_one_group_per_slot():
    assert broker.batch_cap(16, 2) == 8 and broker.batch_cap(17, 2) == 9 and broker.batch_cap(256, 2) == 128
    assert broker.batch_cap(1, 2) == 1 and broker.batch_cap(0, 2) == 1 and broker.batch_cap(5, 1) == 5


class _OverlapContext(_FakeContext):
    """Counts how many device calls are inside the 'device' at the same time."""
    lock = threading.Lock()
    inside = 0
    most = 0

    def denoise_f32_host_v(self, *a, **k):
        with _OverlapContext.lock:
            _OverlapContext.inside += 1
            _OverlapContext.most = max(_OverlapContext.most, _OverlapContext.inside)
        try:
            super().denoise_f32_host_v(*a, **k)
        finally:
            with _OverlapContext.lock:
                _OverlapContext.inside -= 1


def test_two_calls_in_flight_and_one_slot(monkeypatch, tmp_path):
    """slots=2: the clients are cut into two groups whose calls overlap (one group's workers do their host
    work while the other group's call runs); slots=1 is the lockstep form.  Same answers either way."""
    for slots, want_overlap in ((2, True), (1, False)):
        d = tmp_path / f"s{slots}"
        d.mkdir()
        monkeypatch.setenv(broker.ENV_DIR, str(d))
        monkeypatch.setenv("EXABM4D_DEVICE", "0")
        _OverlapContext.most = 0
        fake = _OverlapContext()
        monkeypatch.setattr(_native, "context", lambda device=None: fake)
        monkeypatch.setattr(_native, "new_context", lambda device=None: fake)
        broker._authkey(0, create=True)
        t = threading.Thread(target=broker.serve, kwargs=dict(device=0, idle=1.0, linger=0.01, slots=slots),
                             daemon=True)
        t.start()
        for _ in range(200):
            if os.path.exists(broker.socket_path(0)):
                break
            time.sleep(0.01)
        with multiprocessing.get_context("fork").Pool(6) as pool:
            assert all(pool.map(_worker, [(s, 6) for s in range(6)], chunksize=1))
        t.join(timeout=10)
        assert not t.is_alive()
        assert sum(fake.batches) == 6 * 6 + 6 * 2
        assert max(fake.batches) <= (6 if slots == 2 else 12)      # requests per call <= ceil(6 / slots), two volumes each at most
        assert (_OverlapContext.most >= 2) == want_overlap


def test_broker_reports_errors_instead_of_hanging(monkeypatch, tmp_path):
    monkeypatch.setenv(broker.ENV_DIR, str(tmp_path))

    class Failing:
        def denoise_f32_host_v(self, *a, **k):
            raise ValueError("sigma must be > 0")

    monkeypatch.setattr(_native, "context", lambda device=None: Failing())
    monkeypatch.setattr(_native, "new_context", lambda device=None: Failing())
    broker._authkey(0, create=True)
    t = threading.Thread(target=broker.serve, kwargs=dict(device=0, idle=0.5, linger=0.0), daemon=True)
    t.start()
    for _ in range(200):
        if os.path.exists(broker.socket_path(0)):
            break
        time.sleep(0.01)
    c = broker._Client(0, start_timeout=5.0)
    with pytest.raises(broker.BrokerError, match="sigma"):
        c.denoise(np.zeros((8, 8, 8), np.float32), -1.0, (8, 4, 11, 16, 2.7, 3.0, 0.6, 2.0), 2, None)
    c.close()
    t.join(timeout=10)
    assert not t.is_alive()
