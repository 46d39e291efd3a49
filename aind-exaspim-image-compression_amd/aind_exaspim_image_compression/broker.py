"""One GPU-owner process per device that turns many single-patch ``bm4d()`` calls into batched ones.

Why: the only place the reference runs BM4D at scale is a ``ProcessPoolExecutor(num_workers=None -> all
CPUs)`` of forked workers, each calling ``bm4d(raw, sigma)`` on ONE 64^3 patch at a time
(reference scripts/precompute.py:215-228, machine_learning/data_handling.py:332, :1325-1330).  Dropped in
as it is, every worker would open its own HIP context and launch 3 375 reference blocks at 256 CUs -- and
W contexts on a device cost memory and context switches.  With the broker the workers stay what they are
(plain forked Python processes that never touch the GPU): ``bm4d()`` copies the patch into a shared-memory
segment, sends a few bytes over a UNIX socket and sleeps; the broker -- the only process that owns the
device -- collects what is pending, runs ONE ``exabm4d_denoise_f32_host`` call per (shape, sigma, profile)
group and wakes the callers.  Every volume of a batched call gets its own fixed-point unit,
so a patch's result does not depend on what it was batched with: identical to the direct call, bit for bit.

Use: ``EXABM4D_BROKER=1 python scripts/precompute.py`` (no code change: the first worker that calls
``bm4d()`` starts the broker for its device, the others find it), or ``broker.enable()`` /
``broker.start(devices)`` from the parent before it creates its pool.  One broker per device; a worker
picks its device as ``_native.default_device()`` does (worker index mod device count), so an 8-GPU node
runs eight brokers.  A broker exits when it has had no client for ``idle`` seconds.

Protocol (``multiprocessing.connection`` over AF_UNIX, authkey from a 0600 key file next to the socket):
client -> ``("denoise", shm_name, shape, sigma, params_tuple, stages, clip)``; the fp32 data travels in the
named ``multiprocessing.shared_memory`` segment, which the broker overwrites with the result before it
answers ``("ok",)`` or ``("error", message)``.
"""
import argparse
import atexit
import fcntl
import hashlib
import os
import subprocess
import sys
import tempfile
import time
from multiprocessing import connection, shared_memory

import numpy as np

ENV_ENABLE = "EXABM4D_BROKER"          # "1": bm4d() / denoise_patches() go through the device's broker
ENV_DIR = "EXABM4D_BROKER_DIR"         # where sockets, key and log files live (default: the temp directory)
_enabled = None
_clients = {}                          # (pid, device) -> _Client


def _dir():
    return os.environ.get(ENV_DIR) or tempfile.gettempdir()


def socket_path(device):
    return os.path.join(_dir(), f"exabm4d-broker-{os.getuid()}-{int(device)}.sock")


def _key_path(device):
    return socket_path(device) + ".key"


def _authkey(device, create=False):
    path = _key_path(device)
    if create:
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(os.urandom(32))
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).digest()


def enabled():
    return _enabled if _enabled is not None else os.environ.get(ENV_ENABLE, "") == "1"


def enable(on=True):
    """Route this process's (and its future children's) ``bm4d()`` / ``denoise_patches()`` calls through the
    brokers.  Call it before creating the worker pool; nothing touches the GPU here."""
    global _enabled
    _enabled = bool(on)


# ---- batching policy (host logic; tests/test_broker.py) -------------------------------------------------
def plan_batches(pending, max_voxels=1 << 30):
    """``pending``: list of (request id, key, voxels) in arrival order, key = everything that must be equal
    inside one device call (shape, sigma, profile, stages, clip).  -> list of lists of request ids: one device
    call each, arrival order kept inside a group, groups in order of their first request, a group split
    where it would exceed ``max_voxels`` (the scratch of a call is ~27 bytes per voxel)."""
    groups, order = {}, []
    for rid, key, vox in pending:
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append((rid, vox))
    calls = []
    for key in order:
        cur, tot = [], 0
        for rid, vox in groups[key]:
            if cur and tot + vox > max_voxels:
                calls.append(cur)
                cur, tot = [], 0
            cur.append(rid)
            tot += vox
        calls.append(cur)
    return calls


# ---- client -----------------------------------------------------------------------------------------------
class BrokerError(RuntimeError):
    pass


class _Client:
    def __init__(self, device, start_timeout=180.0):
        self.device = int(device)
        self.conn = _connect_or_start(self.device, start_timeout)
        self.shm = None
        atexit.register(self.close)

    def _segment(self, nbytes):
        if self.shm is None or self.shm.size < nbytes:
            if self.shm is not None:
                self.shm.close()
                self.shm.unlink()
            self.shm = shared_memory.SharedMemory(create=True, size=int(nbytes))
        return self.shm

    def denoise(self, arr, sigma, params_tuple, stages, clip):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        shm = self._segment(arr.nbytes)
        view = np.ndarray(arr.shape, dtype=np.float32, buffer=shm.buf)
        view[...] = arr
        self.conn.send(("denoise", shm.name, tuple(arr.shape), float(sigma), tuple(params_tuple), int(stages),
                        None if clip is None else (float(clip[0]), float(clip[1]))))
        reply = self.conn.recv()
        if reply[0] != "ok":
            raise BrokerError(reply[1] if len(reply) > 1 else "broker failed")
        return view.copy()

    def close(self):
        try:
            self.conn.close()
        except Exception:
            pass
        if self.shm is not None:
            try:
                self.shm.close()
                self.shm.unlink()
            except Exception:
                pass
            self.shm = None


def _try_connect(device):
    try:
        return connection.Client(socket_path(device), family="AF_UNIX", authkey=_authkey(device))
    except (FileNotFoundError, ConnectionRefusedError, OSError, connection.AuthenticationError):
        return None


def _spawn(device):
    """Start the broker of ``device`` as a detached child (its own session: it outlives the worker that
    happened to start it and leaves when it has been idle)."""
    env = dict(os.environ)
    env.pop(ENV_ENABLE, None)                              # the broker itself talks to the GPU directly
    pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = pkg_root + os.pathsep + env.get("PYTHONPATH", "")
    log = open(socket_path(device) + ".log", "ab")
    return subprocess.Popen([sys.executable, "-m", "aind_exaspim_image_compression.broker", "--device",
                             str(int(device))], env=env, stdin=subprocess.DEVNULL, stdout=log, stderr=log,
                            start_new_session=True, close_fds=True)


def _connect_or_start(device, timeout):
    c = _try_connect(device)
    if c is not None:
        return c
    lock = open(socket_path(device) + ".lock", "w")
    try:
        fcntl.flock(lock, fcntl.LOCK_EX)                   # one starter; the others wait here and then connect
        c = _try_connect(device)
        if c is not None:
            return c
        try:
            os.unlink(socket_path(device))                 # a stale socket of a broker that died
        except FileNotFoundError:
            pass
        _authkey(device, create=True)
        proc = _spawn(device)
        t0 = time.time()
        while time.time() - t0 < timeout:
            c = _try_connect(device)
            if c is not None:
                return c
            if proc.poll() is not None:
                raise BrokerError(f"the broker of device {device} exited with code {proc.returncode}; see "
                                  f"{socket_path(device)}.log")
            time.sleep(0.05)
        raise BrokerError(f"the broker of device {device} did not come up within {timeout:.0f} s")
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def client(device):
    key = (os.getpid(), int(device))
    c = _clients.get(key)
    if c is None:
        c = _clients[key] = _Client(device)
    return c


def start(devices=None):
    """Start (or find) the brokers of ``devices`` (default: all visible) from the calling process WITHOUT
    touching the GPU here; returns the device list.  Optional -- workers start a missing broker themselves."""
    from aind_exaspim_image_compression import _native
    if devices is None or devices == "all":
        devices = list(range(max(1, _native.device_count_no_init())))
    for d in devices:
        _connect_or_start(int(d), 180.0).close()
    return list(devices)


def denoise(arr, sigma, params, stages, clip, device=None):
    """What ``bm4d.bm4d`` / ``denoise_patches`` call when the broker is enabled: one fp32 volume [Z, Y, X] or a
    batch [N, Z, Y, X] through the broker of ``device`` (default: this worker's device)."""
    from aind_exaspim_image_compression import _native
    if device is None:
        device = _native.default_device()
    ptuple = (params.block, params.step, params.search, params.max_group, params.lambda_ht, params.c_match_ht,
              params.c_match_wie, params.kaiser_beta)
    return client(device).denoise(arr, sigma, ptuple, stages, clip)


# ---- server -----------------------------------------------------------------------------------------------
def serve(device, idle=10.0, linger=0.002, max_voxels=1 << 30):
    from aind_exaspim_image_compression import _native
    path = socket_path(device)
    listener = connection.Listener(path, family="AF_UNIX", authkey=_authkey(device))
    os.chmod(path, 0o600)
    ctx = _native.context(int(device))
    conns, pending = [], {}               # pending: conn -> request
    last_busy = time.time()
    stats = {"requests": 0, "calls": 0}
    try:
        while True:
            ready = connection.wait([listener._listener._socket] + conns, timeout=0.25 if not pending else 0.0)
            for r in ready:
                if r is listener._listener._socket:
                    try:
                        conns.append(listener.accept())
                    except (connection.AuthenticationError, OSError):
                        pass
                    continue
                try:
                    msg = r.recv()
                except (EOFError, OSError):
                    conns.remove(r)
                    pending.pop(r, None)
                    continue
                if msg[0] == "denoise":
                    pending[r] = msg
                elif msg[0] == "stats":
                    r.send(("ok", dict(stats)))
                elif msg[0] == "shutdown":
                    r.send(("ok",))
                    return
            if conns or pending:
                last_busy = time.time()
            elif time.time() - last_busy > idle:
                return
            if not pending:
                continue
            # every client has at most one request outstanding: once all of them wait, nothing more can
            # arrive; otherwise give the stragglers `linger` seconds
            if len(pending) < len(conns):
                more = connection.wait(conns, timeout=linger)
                if more:
                    continue
            reqs = list(pending.items())
            pending.clear()
            keyed = [(i, (m[2][-3:], m[3], m[4], m[5], m[6]), int(np.prod(m[2]))) for i, (_, m) in enumerate(reqs)]
            for call in plan_batches(keyed, max_voxels):
                _run_call(ctx, [reqs[i] for i in call], _native)
                stats["calls"] += 1
                stats["requests"] += len(call)
    finally:
        listener.close()
        for p in (path,):
            try:
                os.unlink(p)
            except FileNotFoundError:
                pass


def _run_call(ctx, reqs, _native):
    """One device call for requests that share shape[-3:], sigma, profile, stages and clip."""
    segs, views = [], []
    try:
        for _, m in reqs:
            seg = shared_memory.SharedMemory(name=m[1])
            try:        # the segment is the client's: keep this process's resource tracker out of it
                from multiprocessing import resource_tracker
                resource_tracker.unregister(seg._name, "shared_memory")
            except Exception:
                pass
            segs.append(seg)
            shape = tuple(m[2])
            views.append(np.ndarray(shape if len(shape) == 4 else (1,) + shape, dtype=np.float32, buffer=seg.buf))
        _, m0 = reqs[0]
        block, step, search, max_group, lam, c_ht, c_wie, beta = m0[4]
        params = _native.default_params(block=block, step=step, search=search, max_group=max_group,
                                        lambda_ht=lam, c_match_ht=c_ht, c_match_wie=c_wie, kaiser_beta=beta)
        batch = views[0] if len(views) == 1 else np.concatenate(views, axis=0)
        out = ctx.denoise_f32_host(batch, m0[3], params=params, stages=m0[5], clip=m0[6])
        at = 0
        for v in views:
            v[...] = out[at:at + v.shape[0]]
            at += v.shape[0]
        for conn, _ in reqs:
            conn.send(("ok",))
    except Exception as e:                               # the callers must not hang on a failed call
        for conn, _ in reqs:
            try:
                conn.send(("error", f"{type(e).__name__}: {e}"))
            except Exception:
                pass
    finally:
        del views
        for seg in segs:
            try:
                seg.close()
            except Exception:
                pass


def main(argv=None):
    ap = argparse.ArgumentParser(description="GPU-owner process of one device (see the module docstring)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--idle", type=float, default=float(os.environ.get("EXABM4D_BROKER_IDLE", "10")))
    ap.add_argument("--linger", type=float, default=float(os.environ.get("EXABM4D_BROKER_LINGER", "0.002")))
    a = ap.parse_args(argv)
    serve(a.device, idle=a.idle, linger=a.linger)


if __name__ == "__main__":
    main()
