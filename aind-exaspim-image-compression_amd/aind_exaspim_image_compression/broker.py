"""One GPU-owner process per device that turns many single-patch ``bm4d()`` calls into batched ones.

Why: the only place the reference runs BM4D at scale is a ``ProcessPoolExecutor(num_workers=None -> all
CPUs)`` of forked workers, each calling ``bm4d(raw, sigma)`` on ONE 64^3 patch at a time
(reference scripts/precompute.py:215-228, machine_learning/data_handling.py:332, :1325-1330).  Dropped in
as it is, every worker would open its own HIP context and launch 3 375 reference blocks at 256 CUs -- and
W contexts on a device cost memory and context switches.  With the broker the workers stay what they are
(plain forked Python processes that never touch the GPU): ``bm4d()`` copies the patch into a shared-memory
segment, sends a few bytes over a UNIX socket and sleeps; the broker -- the only process that owns the
device -- collects what is pending, runs ONE ``exabm4d_denoise_f32_host_v`` call per (shape, sigma, profile)
group, in place on the callers' segments, and wakes the callers.  Every volume of a batched call gets its own fixed-point unit,
so a patch's result does not depend on what it was batched with: identical to the direct call, bit for bit.

Use: ``EXABM4D_BROKER=1 python scripts/precompute.py`` (no code change: the first worker that calls
``bm4d()`` starts the broker for its device, the others find it), or ``broker.enable()`` /
``broker.start(devices)`` from the parent before it creates its pool.  One broker per device; a worker
picks its device as ``_native.default_device()`` does (worker index mod device count), so an 8-GPU node
runs eight brokers.  A broker exits when it has had no client for ``idle`` seconds.

Several calls in flight (``slots``, default 4: a thread with its own context -- stream, scratch -- each).  A
pool of synchronous workers would otherwise move in lockstep: all of them wait, one call runs, all of them do
their host work while the device idles; and a call of 16 patches is latency, not throughput (3.7 ms of which
1.5 are two block-matching launches that take as long for one patch).  The clients are cut into ``slots``
groups (``batch_cap``); a call starts as soon as every client that is not inside a call has asked, so the
groups drift apart and overlap.  Measured on one MI355X, 16 workers, 1000 patches of 64^3
(``tools/bench_workers.py``): one slot 0.425 s, two 0.37-0.40, four 0.31-0.33, six 0.33, eight 0.43; the same
patches in one ``denoise_patches`` call 0.244-0.265 s.

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
import queue
import tempfile
import threading
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


def stats(device=0):
    """Counters of a running broker: requests, device calls, volumes, seconds spent collecting a batch, inside
    the device calls and answering (None when no broker of ``device`` is up)."""
    c = _try_connect(device)
    if c is None:
        return None
    try:
        c.send(("stats",))
        reply = c.recv()
        return reply[1] if reply[0] == "ok" else None
    finally:
        c.close()


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
class _Segments:
    """The callers' shared-memory segments as the broker sees them: attached once per (connection, name),
    page-locked for the device (``exabm4d_host_register``: the copies of every later call are DMA transfers)
    and kept -- a worker reuses its segment for every patch."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.by_conn = {}                  # conn -> (name, SharedMemory, base address, registered)

    def address(self, conn, name, nbytes):
        cur = self.by_conn.get(conn)
        if cur is None or cur[0] != name:
            self.drop(conn)
            seg = shared_memory.SharedMemory(name=name)
            try:        # the segment is the client's: keep this process's resource tracker out of it
                from multiprocessing import resource_tracker
                resource_tracker.unregister(seg._name, "shared_memory")
            except Exception:
                pass
            base = np.frombuffer(seg.buf, dtype=np.uint8).ctypes.data
            registered = False
            if hasattr(self.ctx, "host_register"):
                try:
                    self.ctx.host_register(base, seg.size)
                    registered = True
                except Exception:          # pageable copies work as well, only slower
                    pass
            cur = self.by_conn[conn] = (name, seg, base, registered)
        if cur[1].size < nbytes:
            raise ValueError("the request is larger than its shared-memory segment")
        return cur[2]

    def drop(self, conn):
        cur = self.by_conn.pop(conn, None)
        if cur is not None:
            if cur[3]:
                try:
                    self.ctx.host_unregister(cur[2])
                except Exception:
                    pass
            try:
                cur[1].close()
            except Exception:      # (BufferError while a view is alive: the mapping goes with the process)
                pass

    def close(self):
        for conn in list(self.by_conn):
            self.drop(conn)


class _Slot(threading.Thread):
    """One device call in flight: a thread with its own context (stream, scratch).  The main loop hands it a
    job and learns of its end through ``done`` + one byte on the wake-up pipe; all socket traffic stays in the
    main loop."""

    def __init__(self, ctx, done, wake_fd):
        super().__init__(daemon=True)
        self.ctx, self.done, self.wake_fd = ctx, done, wake_fd
        self.jobs = queue.Queue()
        self.busy = False

    def run(self):
        while True:
            job = self.jobs.get()
            if job is None:
                return
            reqs, addrs, shape3, sigma, params, stages, clip = job
            t0 = time.perf_counter()
            err = None
            try:
                self.ctx.denoise_f32_host_v(addrs, addrs, shape3, sigma, params=params, stages=stages, clip=clip)
            except Exception as e:                       # the callers must not hang on a failed call
                err = f"{type(e).__name__}: {e}"
            self.done.put((self, reqs, len(addrs), err, time.perf_counter() - t0))
            os.write(self.wake_fd, b"x")


def batch_cap(clients, slots):
    """Largest number of requests one device call takes: with several calls in flight the clients are cut
    into that many groups, so that one group's call runs while the other groups' workers do their host work
    (a pool of synchronous workers would otherwise move in lockstep with the device idle in between)."""
    return max(1, -(-int(clients) // max(1, int(slots))))


def serve(device, idle=10.0, linger=0.002, max_voxels=1 << 30, slots=4):
    from aind_exaspim_image_compression import _native
    path = socket_path(device)
    listener = connection.Listener(path, family="AF_UNIX", authkey=_authkey(device))
    os.chmod(path, 0o600)
    first = _native.context(int(device))
    ctxs = [first] + [_native.new_context(int(device)) for _ in range(max(1, int(slots)) - 1)]
    wake_r, wake_w = os.pipe()
    done = queue.Queue()
    pool = [_Slot(c, done, wake_w) for c in ctxs]
    for sl in pool:
        sl.start()
    lsock = listener._listener._socket
    conns, pending, inflight = [], {}, set()      # pending: conn -> request (arrival order); inflight: conns
    segments = _Segments(_native.new_context(int(device)))     # registrations: a context no call runs on
    last_busy = time.time()
    stats = {"requests": 0, "calls": 0, "volumes": 0, "call_seconds": 0.0, "slots": len(pool), "largest_call": 0}
    first_arrival = None                           # of the oldest request still pending

    def launch(sl, take):
        reqs = [(c, pending.pop(c)) for c in take]
        _, m0 = reqs[0]
        try:
            shape3 = tuple(int(v) for v in m0[2][-3:])
            vol_bytes = 4 * shape3[0] * shape3[1] * shape3[2]
            addrs = []
            for conn, m in reqs:
                count = int(m[2][0]) if len(m[2]) == 4 else 1
                base = segments.address(conn, m[1], count * vol_bytes)
                addrs.extend(base + k * vol_bytes for k in range(count))
            block, step, search, max_group, lam, c_ht, c_wie, beta = m0[4]
            params = _native.default_params(block=block, step=step, search=search, max_group=max_group,
                                            lambda_ht=lam, c_match_ht=c_ht, c_match_wie=c_wie, kaiser_beta=beta)
        except Exception as e:
            for conn, _ in reqs:
                _reply(conn, ("error", f"{type(e).__name__}: {e}"))
            return
        inflight.update(c for c, _ in reqs)
        sl.busy = True
        sl.jobs.put((reqs, addrs, shape3, m0[3], params, m0[5], m0[6]))

    try:
        while True:
            # sleep until a message, the end of a call (wake-up pipe) or -- requests pending and a slot free --
            # the moment the oldest request has lingered long enough
            if pending and any(not sl.busy for sl in pool):
                timeout = max(0.0, linger - (time.perf_counter() - first_arrival))
            else:
                timeout = 0.25
            ready = connection.wait([lsock, wake_r] + conns, timeout=timeout)
            for r in ready:
                if r is lsock:
                    try:
                        conns.append(listener.accept())
                    except (connection.AuthenticationError, OSError):
                        pass
                    continue
                if r == wake_r:
                    os.read(wake_r, 4096)
                    continue
                try:
                    msg = r.recv()
                except (EOFError, OSError):
                    conns.remove(r)
                    pending.pop(r, None)
                    if r not in inflight:          # (a caller that died mid-call: dropped when its call ends)
                        segments.drop(r)
                    continue
                if msg[0] == "denoise":
                    if not pending:
                        first_arrival = time.perf_counter()
                    pending[r] = msg
                elif msg[0] == "stats":
                    r.send(("ok", dict(stats)))
                elif msg[0] == "shutdown":
                    r.send(("ok",))
                    return
            while True:                            # calls that ended: answer their callers
                try:
                    sl, reqs, nvol, err, dt = done.get_nowait()
                except queue.Empty:
                    break
                sl.busy = False
                for conn, _ in reqs:
                    inflight.discard(conn)
                    _reply(conn, ("ok",) if err is None else ("error", err))
                    if conn not in conns:
                        segments.drop(conn)
                stats["calls"] += 1
                stats["requests"] += len(reqs)
                stats["volumes"] += nvol
                stats["call_seconds"] += dt
                stats["largest_call"] = max(stats["largest_call"], nvol)
            if conns or pending or inflight:
                last_busy = time.time()
            elif time.time() - last_busy > idle:
                return
            # A call starts when a slot is free and either nothing more can arrive -- every client has at most
            # one request outstanding, so once all clients that are not inside a call have asked, waiting is
            # pointless -- or the oldest request has waited `linger`.  One call takes at most batch_cap()
            # requests, same (shape, sigma, profile) only.
            while pending:
                free = [sl for sl in pool if not sl.busy]
                if not free:
                    break
                could_still_ask = sum(1 for c in conns if c not in pending and c not in inflight)
                if could_still_ask and time.perf_counter() - first_arrival < linger:
                    break
                order = list(pending)
                keyed = [(c, (pending[c][2][-3:], pending[c][3], pending[c][4], pending[c][5], pending[c][6]),
                          int(np.prod(pending[c][2]))) for c in order]
                call = plan_batches(keyed, max_voxels)[0][:batch_cap(len(conns), len(pool))]
                launch(free[0], call)
                first_arrival = time.perf_counter()
    finally:
        listener.close()
        for sl in pool:
            sl.jobs.put(None)
        for sl in pool:
            sl.join(timeout=30)
        segments.close()
        for fd in (wake_r, wake_w):
            try:
                os.close(fd)
            except OSError:
                pass
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass


def _reply(conn, msg):
    try:
        conn.send(msg)
    except Exception:
        pass


def main(argv=None):
    ap = argparse.ArgumentParser(description="GPU-owner process of one device (see the module docstring)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--idle", type=float, default=float(os.environ.get("EXABM4D_BROKER_IDLE", "10")))
    ap.add_argument("--linger", type=float, default=float(os.environ.get("EXABM4D_BROKER_LINGER", "0.002")))
    ap.add_argument("--slots", type=int, default=int(os.environ.get("EXABM4D_BROKER_SLOTS", "4")),
                    help="device calls in flight (each has its own context)")
    a = ap.parse_args(argv)
    serve(a.device, idle=a.idle, linger=a.linger, slots=a.slots)


if __name__ == "__main__":
    main()
