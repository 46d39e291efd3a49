"""Host <-> device copy rates on the box: pageable hipMemcpy (what exabm4d_memcpy_h2d / d2h do), the cost of
hipHostRegister on a numpy array, and copies from / to the registered array.  One process, 2 GiB."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "aind-exaspim-image-compression_amd")]
from aind_exaspim_image_compression import _native

ctx = _native.context(0)
hip_path = sorted(_native._mapped_hip_runtimes())[0]
hip = ctypes.CDLL(hip_path)
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipDeviceSynchronize.argtypes = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30          # uint16 elements
a = np.random.default_rng(0).integers(0, 65536, n, dtype=np.uint16)
b = np.empty_like(a)
d = ctx.alloc(a.nbytes)
gb = a.nbytes / 1e9


def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); hip.hipDeviceSynchronize(); best = min(best, time.perf_counter() - t0)
    return best


h2d = t(lambda: hip.hipMemcpy(d.ptr, a.ctypes.data, a.nbytes, 1))
d2h = t(lambda: hip.hipMemcpy(b.ctypes.data, d.ptr, a.nbytes, 2))
assert np.array_equal(a, b)
print(f"pageable  H2D {gb / h2d:6.1f} GB/s   D2H {gb / d2h:6.1f} GB/s   ({gb:.2f} GB)", flush=True)
t0 = time.perf_counter(); rc = hip.hipHostRegister(a.ctypes.data, a.nbytes, 0); reg = time.perf_counter() - t0
t0 = time.perf_counter(); rc2 = hip.hipHostRegister(b.ctypes.data, b.nbytes, 0); reg2 = time.perf_counter() - t0
print(f"hipHostRegister rc {rc} {rc2}: {gb / reg:6.1f} GB/s (touched array), {gb / reg2:6.1f} GB/s (second)", flush=True)
h2d = t(lambda: hip.hipMemcpy(d.ptr, a.ctypes.data, a.nbytes, 1))
d2h = t(lambda: hip.hipMemcpy(b.ctypes.data, d.ptr, a.nbytes, 2))
print(f"registered H2D {gb / h2d:6.1f} GB/s   D2H {gb / d2h:6.1f} GB/s", flush=True)
t0 = time.perf_counter(); hip.hipHostUnregister(a.ctypes.data); hip.hipHostUnregister(b.ctypes.data)
print(f"unregister both: {time.perf_counter() - t0:.3f} s", flush=True)
# both directions at once from two threads (registered again), as the streamed driver would
import threading
hip.hipHostRegister(a.ctypes.data, a.nbytes, 0); hip.hipHostRegister(b.ctypes.data, b.nbytes, 0)
d2 = ctx.alloc(a.nbytes)
def both():
    th = threading.Thread(target=lambda: hip.hipMemcpy(d.ptr, a.ctypes.data, a.nbytes, 1))
    th.start(); hip.hipMemcpy(b.ctypes.data, d2.ptr, a.nbytes, 2); th.join()
tb = t(both)
print(f"registered, both directions together: {2 * gb / tb:6.1f} GB/s in total", flush=True)
