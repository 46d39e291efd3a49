"""ctypes binding of ``libexabm4d.so`` (the C-ABI declared in ``include/exabm4d.h``).

There is no CPU fallback: if the shared library is missing, or no MI355X is visible when a
compute entry point is called, this module raises.  Contexts are per (process id, device) and are
created lazily, *after* any ``fork()`` -- the reference calls ``bm4d`` from forked
``ProcessPoolExecutor`` workers (reference ``scripts/precompute.py:215``).
"""
import ctypes
import importlib.util
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_CANDIDATES = [
    os.environ.get("EXABM4D_LIB", ""),
    os.path.join(os.path.dirname(_HERE), "csrc", "libexabm4d.so"),
]

c_f32p = ctypes.POINTER(ctypes.c_float)
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_vp = ctypes.c_void_p

KEY_EMPTY = 0xFFFFFFFF
DATA_EXP_AUTO = -2 ** 31    # exabm4d.h EXABM4D_DATA_EXP_AUTO: E per volume from the data (fp32 entry points)
DATA_EXP_U16 = 17           # exabm4d.h EXABM4D_DATA_EXP_U16: the uint16 pipelines' fixed E


class Params(ctypes.Structure):
    """``exabm4d_params`` (include/exabm4d.h)."""

    _fields_ = [
        ("size", ctypes.c_uint32),
        ("block", ctypes.c_int32),
        ("step", ctypes.c_int32),
        ("search", ctypes.c_int32),
        ("max_group", ctypes.c_int32),
        ("lambda_ht", ctypes.c_float),
        ("c_match_ht", ctypes.c_float),
        ("c_match_wie", ctypes.c_float),
        ("kaiser_beta", ctypes.c_float),
    ]


class Transform(ctypes.Structure):
    """``exabm4d_transform`` (include/exabm4d.h)."""

    _fields_ = [
        ("size", ctypes.c_uint32),
        ("kind", ctypes.c_int32),
        ("wrapped", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("wrap_offset", ctypes.c_double),
        ("max_count", ctypes.c_double),
        ("offset", ctypes.c_double),
        ("scale", ctypes.c_double),
        ("norm", ctypes.c_double),
        ("gain", ctypes.c_double),
        ("read_noise", ctypes.c_double),
        ("c_inv", ctypes.c_double),
        ("mn", ctypes.c_double),
        ("mx", ctypes.c_double),
        ("clip", ctypes.c_double),
    ]


# name -> (restype, argtypes); every symbol include/exabm4d.h declares
_CTX = c_vp
_PP = ctypes.POINTER(Params)
_TP = ctypes.POINTER(Transform)
_I, _F, _SZ = ctypes.c_int, ctypes.c_float, ctypes.c_size_t
SIGNATURES = {
    "exabm4d_version": (_I, []),
    "exabm4d_last_error": (ctypes.c_char_p, [_CTX]),
    "exabm4d_device_count": (_I, []),
    "exabm4d_create": (_I, [_I, ctypes.POINTER(_CTX)]),
    "exabm4d_destroy": (_I, [_CTX]),
    "exabm4d_set_stream": (_I, [_CTX, c_vp]),
    "exabm4d_reset_stream": (_I, [_CTX]),
    "exabm4d_sync": (_I, [_CTX]),
    "exabm4d_default_params": (_I, [_PP]),
    "exabm4d_set_option": (_I, [_CTX, ctypes.c_char_p, _I]),
    "exabm4d_profile_read": (_I, [_CTX, c_f32p, _I]),
    "exabm4d_malloc": (_I, [_CTX, _SZ, ctypes.POINTER(c_vp)]),
    "exabm4d_free": (_I, [_CTX, c_vp]),
    "exabm4d_memcpy_h2d": (_I, [_CTX, c_vp, c_vp, _SZ]),
    "exabm4d_memcpy_d2h": (_I, [_CTX, c_vp, c_vp, _SZ]),
    "exabm4d_memset": (_I, [_CTX, c_vp, _I, _SZ]),
    "exabm4d_event_create": (_I, [_CTX, ctypes.POINTER(c_vp)]),
    "exabm4d_event_destroy": (_I, [_CTX, c_vp]),
    "exabm4d_event_record": (_I, [_CTX, c_vp]),
    "exabm4d_event_elapsed_ms": (_I, [_CTX, c_vp, c_vp, c_f32p]),
    "exabm4d_grid_count": (_I, [_I]),
    "exabm4d_grid_positions": (_I, [_I, c_i32p]),
    "exabm4d_tables": (_I, [_PP, c_f32p, c_f32p]),
    "exabm4d_scratch_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "exabm4d_blockmatch_dev": (_I, [_CTX, c_vp, _I, _I, _I, _I, _F, _F, _PP, c_vp]),
    "exabm4d_blockmatch_u16_dev": (_I, [_CTX, c_vp, _I, _I, _I, _I, _F, _F, _PP, c_vp]),
    "exabm4d_match_decode": (_I, [c_u32p, _I, _I, _I, _I, _I, c_i64p, c_f32p,
                                  ctypes.POINTER(_I)]),
    "exabm4d_stage_dev": (_I, [_CTX, c_vp, c_vp, c_vp, _I, _I, _I, _I, _F, _PP, _I, c_vp, c_vp]),
    "exabm4d_normalize_dev": (_I, [_CTX, c_vp, c_vp, c_vp, _SZ, _F, _F]),
    "exabm4d_counts_from_u16_dev": (_I, [_CTX, c_vp, c_vp, _SZ, _F]),
    "exabm4d_round_counts_f32_dev": (_I, [_CTX, c_vp, c_vp, _SZ, _F]),
    "exabm4d_normalize_u16_dev": (_I, [_CTX, c_vp, c_vp, c_vp, _SZ, _F]),
    "exabm4d_denoise_f32_dev": (_I, [_CTX, c_vp, c_vp, _I, _I, _I, _I, _F, _PP, _I, _F, _F]),
    "exabm4d_denoise_u16_dev": (_I, [_CTX, c_vp, c_vp, _I, _I, _I, _I, _F, _F, _PP, _I]),
    "exabm4d_denoise_chunked_u16_dev": (_I, [_CTX, c_vp, c_vp, _I, _I, _I, _I, _I, _I, _I, _F, _F, _PP,
                                             _I]),
    "exabm4d_blockmatch_plan": (_I, [_CTX, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64)]),
    "exabm4d_denoise_chunked_u16_host": (_I, [_CTX, c_vp, c_vp, _I, _I, _I, _I, _I, _F, _F, _PP, _I]),
    "exabm4d_denoise_f32_host": (_I, [_CTX, c_vp, c_vp, _I, _I, _I, _I, _F, _PP, _I, _F, _F]),
    "exabm4d_denoise_f32_host_v": (_I, [_CTX, c_vp, c_vp, _I, _I, _I, _I, _F, _PP, _I, _F, _F]),
    "exabm4d_groupnorm_workspace_bytes": (_SZ, [_I, _SZ, _I, _I]),
    "exabm4d_groupnorm_lrelu_ndhwc_dev": (_I, [_CTX, c_vp, c_vp, c_vp, _I, _SZ, _I, _I, c_vp, c_vp, _F, _F, c_vp, _SZ,
                                                c_vp]),
    "exabm4d_maxpool2_ndhwc_dev": (_I, [_CTX, c_vp, c_vp, c_vp, _I, _I, _I, _I, _I]),
    "exabm4d_upsample2_trilinear_ndhwc_dev": (_I, [_CTX, c_vp, c_vp, c_vp, _I, _I, _I, _I, _I]),
    "exabm4d_host_register": (_I, [_CTX, c_vp, ctypes.c_size_t]),
    "exabm4d_host_unregister": (_I, [_CTX, c_vp]),
    "exabm4d_transform_forward_u16_dev": (_I, [_CTX, _TP, c_vp, c_vp, _SZ]),
    "exabm4d_transform_forward_f32_dev": (_I, [_CTX, _TP, c_vp, c_vp, _SZ]),
    "exabm4d_transform_inverse_u16_dev": (_I, [_CTX, _TP, c_vp, c_vp, _SZ]),
    "exabm4d_transform_inverse_f32_dev": (_I, [_CTX, _TP, c_vp, c_vp, _SZ]),
    "exabm4d_tile_gather_dev": (_I, [_CTX, c_vp, _I, _I, _I, c_i32p, _I, _I, c_vp]),
    "exabm4d_tile_accumulate_dev": (_I, [_CTX, c_vp, c_i32p, _I, _I, _I, c_vp, c_vp, _I, _I, _I]),
    "exabm4d_tile_finalize_u16_dev": (_I, [_CTX, _TP, c_vp, c_vp, c_vp, _SZ]),
    "exabm4d_chunk_byte_histograms_dev": (_I, [_CTX, c_vp, _I, _I, _I, _I, _I, _I, c_vp]),
    "exabm4d_dctq_forward_dev": (_I, [_CTX, c_vp, _I, _I, _I, ctypes.c_float, c_vp]),
    "exabm4d_dctq_inverse_dev": (_I, [_CTX, c_vp, _I, _I, _I, ctypes.c_float, c_vp]),
    "exabm4d_i32_symbol_histogram_dev": (_I, [_CTX, c_vp, _SZ, c_vp]),
    "exabm4d_comm_unique_id": (_I, [c_vp]),
    "exabm4d_comm_create": (_I, [_CTX, _I, _I, c_vp, ctypes.POINTER(c_vp)]),
    "exabm4d_comm_destroy": (_I, [c_vp]),
    "exabm4d_halo_exchange_dev": (_I, [_CTX, c_vp, _I, c_vp, c_vp, _SZ, _I, c_vp, c_vp, _SZ]),
    "exabm4d_comm_max_f64_host": (_I, [_CTX, c_vp, ctypes.POINTER(ctypes.c_double)]),
    "exabm4d_codec_chunk_bound": (_SZ, [_SZ, _I]),
    "exabm4d_codec_volume_bound": (_SZ, [_I, _I, _I, _I, _I, _I, _I]),
    "exabm4d_codec_encode_dev": (_I, [_CTX, c_vp, _I, _I, _I, _I, _I, _I, _I, _I, c_vp, _SZ, c_vp, c_vp,
                                      c_vp]),
    "exabm4d_codec_decode_dev": (_I, [_CTX, c_vp, _SZ, c_vp, _I, _I, _I, _I, _I, _I, _I, c_vp]),
    "exabm4d_u16_histogram_dev": (_I, [_CTX, c_vp, _SZ, c_vp]),
    "exabm4d_key_histogram_dev": (_I, [_CTX, c_vp, _I, _SZ, _I, ctypes.c_double, _I,
                                       ctypes.c_uint64, c_vp]),
    "exabm4d_minmax_dev": (_I, [_CTX, c_vp, _I, _SZ, c_vp]),
    "exabm4d_masked_error_stats_dev": (_I, [_CTX, c_vp, _I, c_vp, _I, c_vp, _SZ,
                                            ctypes.c_double, c_vp]),
    "exabm4d_ssim3d_dev": (_I, [_CTX, c_vp, c_vp, _I, _I, _I, _I, _I, ctypes.c_double,
                                ctypes.c_double, c_vp]),
}

_lib = None
_lib_lock = threading.Lock()


class NativeError(RuntimeError):
    """libexabm4d.so is missing, or a call into it failed."""


def library_path():
    for p in _LIB_CANDIDATES:
        if p and os.path.exists(p):
            return p
    raise NativeError(
        "libexabm4d.so not found (looked in %s). Build it with `make -C "
        "aind-exaspim-image-compression_amd/csrc` or __graft_entry__.build(); there is no "
        "CPU fallback for the HIP hot path." % [p for p in _LIB_CANDIDATES if p])


def _mapped_hip_runtimes():
    """Real paths of every libamdhip64 image mapped into this process (/proc/self/maps)."""
    seen = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    seen.add(os.path.realpath(line.split(None, 5)[-1].strip()))
    except OSError:
        pass
    return seen


def _adopt_torch_hip_runtime():
    """PyTorch-ROCm wheels carry their own libamdhip64.so (same SONAME as /opt/rocm's).  Two copies
    in one process leave the one loaded second without devices, whichever side it belongs to: the
    reference's callers import ``bm4d`` first and torch later (data_handling.py:12, inference.py),
    so the order is not ours to choose.  Rule: when a torch with a bundled HIP runtime is
    installed, ITS runtime is the process's runtime -- it is mapped here (dlopen only: no HIP
    call, no device initialisation, torch itself is not imported) before libexabm4d.so, whose
    DT_NEEDED ``libamdhip64.so.7`` then binds to the image already loaded under that SONAME.
    ``EXABM4D_SYSTEM_HIP=1`` keeps /opt/rocm's runtime (for processes that never load torch)."""
    if os.environ.get("EXABM4D_SYSTEM_HIP") == "1":
        return None
    if _mapped_hip_runtimes():
        return None                       # somebody (torch, rocprofv3, the caller) already chose
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    return path


def lib():
    """Load the shared library (no GPU needed for this) and bind every symbol."""
    global _lib
    with _lib_lock:
        if _lib is None:
            _adopt_torch_hip_runtime()
            L = ctypes.CDLL(library_path())
            mapped = _mapped_hip_runtimes()
            if len(mapped) > 1:
                raise NativeError(
                    "two HIP runtimes are mapped into this process (%s): the one loaded second has "
                    "no devices.  Load libexabm4d.so and torch against the same libamdhip64 "
                    "(see INTEGRATION.md 1d)." % ", ".join(sorted(mapped)))
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


def default_params(**overrides):
    p = Params()
    rc = lib().exabm4d_default_params(ctypes.byref(p))
    if rc:
        raise NativeError("exabm4d_default_params failed")
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise ValueError(f"unknown BM4D parameter: {k}")
        setattr(p, k, v)
    return p


def _ptr(obj):
    """Device pointer of a DeviceBuffer, a torch tensor, an int, or None."""
    if obj is None:
        return None
    if isinstance(obj, DeviceBuffer):
        return obj.ptr
    if isinstance(obj, int):
        return obj
    if hasattr(obj, "data_ptr"):
        return int(obj.data_ptr())
    raise TypeError(f"cannot take a device pointer from {type(obj)!r}")


class DeviceBuffer:
    """A hipMalloc'd region owned through ``exabm4d_malloc`` (for hosts without torch)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = c_vp()
        ctx._check(lib().exabm4d_malloc(ctx.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.nbytes:
            raise ValueError("host array larger than device buffer")
        self.ctx._check(lib().exabm4d_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data,
                                                 arr.nbytes))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("requested more bytes than the device buffer holds")
        self.ctx._check(lib().exabm4d_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr,
                                                 out.nbytes))
        return out

    def zero(self):
        return self.fill(0)

    def fill(self, byte):
        self.ctx._check(lib().exabm4d_memset(self.ctx.handle, self.ptr, int(byte), self.nbytes))
        return self

    def free(self):
        if self.ptr:
            lib().exabm4d_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One ``exabm4d_ctx`` -- bound to the process that created it and one device."""

    def __init__(self, device=0):
        self.pid = os.getpid()
        self.device = int(device)
        h = c_vp()
        rc = lib().exabm4d_create(self.device, ctypes.byref(h))
        if rc:
            raise NativeError("exabm4d_create(device=%d) failed: %s" % (
                self.device, lib().exabm4d_last_error(None).decode()))
        self.handle = h

    def _check(self, rc):
        if rc:
            msg = lib().exabm4d_last_error(self.handle).decode()
            if rc in (-1, -2):
                raise ValueError(msg)
            raise NativeError(msg)

    # -- memory ----------------------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.nbytes).upload(arr)

    def sync(self):
        self._check(lib().exabm4d_sync(self.handle))

    def set_option(self, name, value):
        self._check(lib().exabm4d_set_option(self.handle, name.encode(), int(value)))

    PHASES = ("counts_from_u16", "zero_acc_1", "blockmatch_ht", "stage_ht", "normalize_basic",
              "zero_acc_2", "blockmatch_wie", "stage_wie", "normalize_out")

    def profile_read(self):
        """Per-phase milliseconds of the last denoise call (needs set_option('profile', 1))."""
        ms = (ctypes.c_float * len(self.PHASES))()
        n = lib().exabm4d_profile_read(self.handle, ms, len(self.PHASES))
        if n < 0:
            self._check(n)
        return {name: float(ms[i]) for i, name in enumerate(self.PHASES[:n])}

    def set_stream(self, hip_stream):
        """Enqueue on an existing HIP stream handle; 0 / None is the HIP null stream (which is
        what torch's default stream is)."""
        self._check(lib().exabm4d_set_stream(self.handle, hip_stream or None))

    def reset_stream(self):
        """Back to the context's private non-blocking stream."""
        self._check(lib().exabm4d_reset_stream(self.handle))

    # -- HIP events on the context's stream -------------------------------------------------
    def event(self):
        e = c_vp()
        self._check(lib().exabm4d_event_create(self.handle, ctypes.byref(e)))
        return e

    def record(self, ev):
        self._check(lib().exabm4d_event_record(self.handle, ev))

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        self._check(lib().exabm4d_event_elapsed_ms(self.handle, a, b, ctypes.byref(ms)))
        return float(ms.value)

    # -- BM4D ---------------------------------------------------------------------------------
    def blockmatch(self, vol, shape, sigma, c_match, keys, params=None, batch=1):
        p = params or default_params()
        nz, ny, nx = shape
        self._check(lib().exabm4d_blockmatch_dev(self.handle, _ptr(vol), nz, ny, nx, batch,
                                                 float(sigma), float(c_match), ctypes.byref(p),
                                                 _ptr(keys)))

    def blockmatch_u16(self, vol, shape, sigma, c_match, keys, params=None, batch=1):
        p = params or default_params()
        nz, ny, nx = shape
        self._check(lib().exabm4d_blockmatch_u16_dev(self.handle, _ptr(vol), nz, ny, nx, batch,
                                                     float(sigma), float(c_match), ctypes.byref(p),
                                                     _ptr(keys)))

    def stage(self, noisy, basic, keys, shape, sigma, num, den, params=None, batch=1, data_exp=None):
        """One collaborative-filtering stage; ``num`` / ``den`` (fp32) are WRITTEN.  ``data_exp``: E of
        the numerator's fixed-point unit; None = per volume from ``noisy`` (the fp32
        pipelines' rule), DATA_EXP_U16 = what the uint16 pipelines use."""
        p = params or default_params()
        nz, ny, nx = shape
        self._check(lib().exabm4d_stage_dev(self.handle, _ptr(noisy), _ptr(basic), _ptr(keys), nz,
                                            ny, nx, batch, float(sigma), ctypes.byref(p),
                                            DATA_EXP_AUTO if data_exp is None else int(data_exp),
                                            _ptr(num), _ptr(den)))

    def normalize(self, num, den, out, n, clip=None):
        lo, hi = (1.0, 0.0) if clip is None else clip
        self._check(lib().exabm4d_normalize_dev(self.handle, _ptr(num), _ptr(den), _ptr(out), n,
                                                float(lo), float(hi)))

    def counts_from_u16(self, src, dst, n, offset):
        """dst (fp32) = (float)src - offset: read_counts, data_handling.py:337-354"""
        self._check(lib().exabm4d_counts_from_u16_dev(self.handle, _ptr(src), _ptr(dst), n, offset))

    def round_counts(self, src, dst, n, offset):
        """dst (fp32) = (float)rint(clamp(src + offset, 0, 65535)) - offset: what stage 2 of the uint16
        pipelines matches on"""
        self._check(lib().exabm4d_round_counts_f32_dev(self.handle, _ptr(src), _ptr(dst), n, offset))

    def normalize_u16(self, num, den, out, n, offset):
        """out (uint16) = rint(clamp(num / den + offset, 0, 65535))"""
        self._check(lib().exabm4d_normalize_u16_dev(self.handle, _ptr(num), _ptr(den), _ptr(out), n,
                                                    offset))

    def denoise_f32(self, src, dst, shape, sigma, params=None, stages=2, clip=None, batch=1):
        p = params or default_params()
        nz, ny, nx = shape
        lo, hi = (1.0, 0.0) if clip is None else clip
        self._check(lib().exabm4d_denoise_f32_dev(self.handle, _ptr(src), _ptr(dst), nz, ny, nx,
                                                  batch, float(sigma), ctypes.byref(p),
                                                  int(stages), float(lo), float(hi)))

    def denoise_u16(self, src, dst, shape, sigma, offset, params=None, stages=2, batch=1):
        p = params or default_params()
        nz, ny, nx = shape
        self._check(lib().exabm4d_denoise_u16_dev(self.handle, _ptr(src), _ptr(dst), nz, ny, nx,
                                                  batch, float(sigma), float(offset),
                                                  ctypes.byref(p), int(stages)))

    def denoise_chunked_u16(self, src, dst, shape, sigma, offset, chunk=256, halo=8, core=None,
                            params=None, stages=2):
        """Chunk-local mode: ``src`` [nz,ny,nx] uint16 on the device, ``dst`` receives the core
        planes ``core = (zc0, zc1)`` (default: all planes)."""
        p = params or default_params()
        nz, ny, nx = shape
        zc0, zc1 = (0, nz) if core is None else core
        self._check(lib().exabm4d_denoise_chunked_u16_dev(
            self.handle, _ptr(src), _ptr(dst), nz, ny, nx, int(zc0), int(zc1), int(chunk),
            int(halo), float(sigma), float(offset), ctypes.byref(p), int(stages)))

    def denoise_chunked_u16_host(self, src, dst, sigma, offset, chunk=256, halo=8, params=None, stages=2):
        """Host uint16 arrays (numpy / memmap, C-contiguous, same 3-D shape) streamed through the device
        one layer of chunks at a time; returns when ``dst`` is complete."""
        p = params or default_params()
        check_host_volume_pair(src, dst)
        nz, ny, nx = src.shape
        self._check(lib().exabm4d_denoise_chunked_u16_host(
            self.handle, src.ctypes.data, dst.ctypes.data, nz, ny, nx, int(chunk), int(halo), float(sigma),
            float(offset), ctypes.byref(p), int(stages)))

    def denoise_f32_host(self, arr, sigma, params=None, stages=2, clip=None):
        """numpy fp32 [N,]Z,Y,X in -> new numpy array out (H2D, kernels, D2H, sync)."""
        p = params or default_params()
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if arr.ndim == 3:
            batch, (nz, ny, nx) = 1, arr.shape
        elif arr.ndim == 4:
            batch, (nz, ny, nx) = arr.shape[0], arr.shape[1:]
        else:
            raise ValueError("expected a 3-D volume or a 4-D batch of volumes")
        out = np.empty_like(arr)
        lo, hi = (1.0, 0.0) if clip is None else clip
        self._check(lib().exabm4d_denoise_f32_host(self.handle, arr.ctypes.data, out.ctypes.data,
                                                   nz, ny, nx, batch, float(sigma),
                                                   ctypes.byref(p), int(stages), float(lo),
                                                   float(hi)))
        return out

    # -- BM4DNet stage ---------------------------------------------------------------------------
    def groupnorm_lrelu_ndhwc(self, stream, x, y, batch, spatial, channels, groups, gamma, beta, eps, slope,
                              workspace, workspace_bytes, conv_bias=None):
        """GroupNorm + LeakyReLU on an NDHWC fp32 tensor (x, y, gamma, beta, workspace: device pointers or
        objects with ``data_ptr()``; ``stream``: the HIP stream handle to run on).  ValueError where the
        fused kernels do not apply (see the header)."""
        self._check(lib().exabm4d_groupnorm_lrelu_ndhwc_dev(
            self.handle, int(stream), _ptr(x), _ptr(y), int(batch), int(spatial), int(channels), int(groups),
            _ptr(gamma) if gamma is not None else None, _ptr(beta) if beta is not None else None,
            float(eps), float(slope), _ptr(workspace), int(workspace_bytes),
            _ptr(conv_bias) if conv_bias is not None else None))

    def maxpool2_ndhwc(self, stream, x, y, batch, d, h, w, channels):
        self._check(lib().exabm4d_maxpool2_ndhwc_dev(self.handle, int(stream), _ptr(x), _ptr(y), int(batch), int(d),
                                                     int(h), int(w), int(channels)))

    def upsample2_trilinear_ndhwc(self, stream, x, y, batch, d, h, w, channels):
        self._check(lib().exabm4d_upsample2_trilinear_ndhwc_dev(self.handle, int(stream), _ptr(x), _ptr(y),
                                                                int(batch), int(d), int(h), int(w), int(channels)))

    def host_register(self, addr, nbytes):
        """Page-lock caller memory that host entry points copy from / to repeatedly (see the header)."""
        self._check(lib().exabm4d_host_register(self.handle, int(addr), int(nbytes)))

    def host_unregister(self, addr):
        self._check(lib().exabm4d_host_unregister(self.handle, int(addr)))

    def denoise_f32_host_v(self, in_addrs, out_addrs, shape, sigma, params=None, stages=2, clip=None):
        """A batch of fp32 volumes of ``shape`` that lie anywhere in host memory: ``in_addrs[i]`` /
        ``out_addrs[i]`` are the addresses of volume i (they may be equal: in place).  The caller keeps the
        memory alive and writable; nothing is allocated or copied on the host side."""
        if len(in_addrs) != len(out_addrs) or not len(in_addrs):
            raise ValueError("one input and one output address per volume")
        p = params or default_params()
        nz, ny, nx = (int(s) for s in shape)
        n = len(in_addrs)
        ins = (ctypes.c_void_p * n)(*[int(a) for a in in_addrs])
        outs = (ctypes.c_void_p * n)(*[int(a) for a in out_addrs])
        lo, hi = (1.0, 0.0) if clip is None else clip
        self._check(lib().exabm4d_denoise_f32_host_v(self.handle, ctypes.cast(ins, c_vp), ctypes.cast(outs, c_vp),
                                                     nz, ny, nx, n, float(sigma), ctypes.byref(p), int(stages),
                                                     float(lo), float(hi)))

    # -- transforms ---------------------------------------------------------------------------
    def transform_forward(self, tf, src, dst, n, src_is_u16):
        fn = (lib().exabm4d_transform_forward_u16_dev if src_is_u16
              else lib().exabm4d_transform_forward_f32_dev)
        self._check(fn(self.handle, ctypes.byref(tf), _ptr(src), _ptr(dst), n))

    def transform_inverse(self, tf, src, dst, n, quantise=True):
        fn = (lib().exabm4d_transform_inverse_u16_dev if quantise
              else lib().exabm4d_transform_inverse_f32_dev)
        self._check(fn(self.handle, ctypes.byref(tf), _ptr(src), _ptr(dst), n))

    # -- tiling -------------------------------------------------------------------------------
    def tile_gather(self, vol, shape, starts, patch, out):
        starts = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, 3)
        nz, ny, nx = shape
        self._check(lib().exabm4d_tile_gather_dev(
            self.handle, _ptr(vol), nz, ny, nx, starts.ctypes.data_as(c_i32p), len(starts),
            int(patch), _ptr(out)))

    def tile_accumulate(self, preds, starts, patch, trim, acc, wgt, shape):
        starts = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, 3)
        nz, ny, nx = shape
        self._check(lib().exabm4d_tile_accumulate_dev(
            self.handle, _ptr(preds), starts.ctypes.data_as(c_i32p), len(starts), int(patch),
            int(trim), _ptr(acc), _ptr(wgt), nz, ny, nx))

    def tile_finalize(self, tf, acc, wgt, out, n):
        self._check(lib().exabm4d_tile_finalize_u16_dev(self.handle, ctypes.byref(tf), _ptr(acc),
                                                        _ptr(wgt), _ptr(out), n))

    def chunk_byte_histograms(self, vol, shape, chunk, hist):
        nz, ny, nx = shape
        self._check(lib().exabm4d_chunk_byte_histograms_dev(self.handle, _ptr(vol), nz, ny, nx,
                                                            int(chunk[0]), int(chunk[1]),
                                                            int(chunk[2]), _ptr(hist)))

    # -- transform quantiser (row f-1) ---------------------------------------------------------
    def dctq_forward(self, vol, shape, q, idx):
        nz, ny, nx = shape
        self._check(lib().exabm4d_dctq_forward_dev(self.handle, _ptr(vol), nz, ny, nx, float(q),
                                                   _ptr(idx)))

    def dctq_inverse(self, idx, shape, q, vol):
        nz, ny, nx = shape
        self._check(lib().exabm4d_dctq_inverse_dev(self.handle, _ptr(idx), nz, ny, nx, float(q),
                                                   _ptr(vol)))

    # -- chunk entropy coder (row f-1) ------------------------------------------------------------
    def codec_encode(self, vol, typesize, shape, chunk, out=None, out_capacity=0, offsets=None,
                     sizes=None, totals=True, version=0):
        """Code every chunk of a device volume.  -> (sum of stream lengths, container bytes) when
        `totals` (synchronises), else None.  ``version``: EXAC format of THIS call (1 | 2; 0 = the
        context's "codec_version" option)."""
        nz, ny, nx = shape
        tot = np.zeros(2, dtype=np.uint64)
        self._check(lib().exabm4d_codec_encode_dev(
            self.handle, _ptr(vol), int(typesize), int(version), nz, ny, nx, int(chunk[0]), int(chunk[1]),
            int(chunk[2]), _ptr(out), int(out_capacity), _ptr(offsets), _ptr(sizes),
            tot.ctypes.data_as(c_vp) if totals else None))
        return (int(tot[0]), int(tot[1])) if totals else None

    def codec_decode(self, data, nbytes, offsets, typesize, shape, chunk, vol):
        """``data``: ``nbytes`` bytes of chunk streams on the device; the decoder never reads
        outside them, whatever ``offsets`` says (malformed containers raise ValueError)."""
        nz, ny, nx = shape
        self._check(lib().exabm4d_codec_decode_dev(
            self.handle, _ptr(data), int(nbytes), _ptr(offsets), int(typesize), nz, ny, nx,
            int(chunk[0]), int(chunk[1]), int(chunk[2]), _ptr(vol)))

    # -- background offset + quality metrics (row f-4); inputs on device, scalars to the host ----
    DTYPES = {np.dtype(np.uint16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}

    def i32_symbol_histogram(self, idx, n):
        hist = np.empty(65536, dtype=np.uint64)
        self._check(lib().exabm4d_i32_symbol_histogram_dev(self.handle, _ptr(idx), n,
                                                           hist.ctypes.data_as(c_vp)))
        return hist

    def u16_histogram(self, vol, n):
        hist = np.empty(65536, dtype=np.uint64)
        self._check(lib().exabm4d_u16_histogram_dev(self.handle, _ptr(vol), n,
                                                    hist.ctypes.data_as(c_vp)))
        return hist

    def key_histogram(self, vol, dtype, n, digit, prefix=0, center=None):
        hist = np.empty(65536, dtype=np.uint64)
        self._check(lib().exabm4d_key_histogram_dev(
            self.handle, _ptr(vol), self.DTYPES[np.dtype(dtype)], n, 0 if center is None else 1,
            0.0 if center is None else float(center), int(digit), int(prefix),
            hist.ctypes.data_as(c_vp)))
        return hist

    def minmax(self, vol, dtype, n):
        out = np.empty(2, dtype=np.float64)
        self._check(lib().exabm4d_minmax_dev(self.handle, _ptr(vol), self.DTYPES[np.dtype(dtype)], n,
                                             out.ctypes.data_as(c_vp)))
        return float(out[0]), float(out[1])

    def masked_error_stats(self, pred, pred_dtype, ref, ref_dtype, mask, n, thr=float("inf")):
        out = np.empty(7, dtype=np.float64)
        self._check(lib().exabm4d_masked_error_stats_dev(
            self.handle, _ptr(pred), self.DTYPES[np.dtype(pred_dtype)], _ptr(ref),
            self.DTYPES[np.dtype(ref_dtype)], _ptr(mask) if mask is not None else None, n,
            float(thr), out.ctypes.data_as(c_vp)))
        return out

    def ssim3d_sum(self, a, b, dtype, shape, window, c1, c2):
        out = np.empty(1, dtype=np.float64)
        nz, ny, nx = shape
        self._check(lib().exabm4d_ssim3d_dev(self.handle, _ptr(a), _ptr(b),
                                             self.DTYPES[np.dtype(dtype)], nz, ny, nx, int(window),
                                             float(c1), float(c2), out.ctypes.data_as(c_vp)))
        return float(out[0])

    def close(self):
        if self.handle and os.getpid() == self.pid:
            lib().exabm4d_destroy(self.handle)
        self.handle = None


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 draws them; every rank passes them to ``Comm``)."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    rc = lib().exabm4d_comm_unique_id(buf)
    if rc:
        raise NativeError(lib().exabm4d_last_error(None).decode())
    return bytes(buf)


class Comm:
    """An RCCL communicator bound to a context's device (exabm4d_comm_*): the halo exchange of the sharded
    modes without torch.distributed.  Creation is collective over the ranks."""

    def __init__(self, ctx, nranks, rank, unique_id):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % COMM_ID_BYTES)
        self.ctx, self.nranks, self.rank = ctx, int(nranks), int(rank)
        h = c_vp()
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        ctx._check(lib().exabm4d_comm_create(ctx.handle, self.nranks, self.rank, buf, ctypes.byref(h)))
        self.handle = h

    def halo_exchange(self, lo_peer, send_lo, recv_lo, bytes_lo, hi_peer, send_hi, recv_hi, bytes_hi):
        """ncclGroupStart; send / recv with each neighbour; ncclGroupEnd -- on the context's stream, device
        pointers (ints, DeviceBuffers or tensors), a peer of -1 skips that side."""
        self.ctx._check(lib().exabm4d_halo_exchange_dev(
            self.ctx.handle, self.handle, int(lo_peer), _ptr(send_lo), _ptr(recv_lo), int(bytes_lo),
            int(hi_peer), _ptr(send_hi), _ptr(recv_hi), int(bytes_hi)))

    def max(self, value):
        """max over the ranks of ``value`` (a float); synchronises the context's stream: barrier + MAX."""
        v = ctypes.c_double(float(value))
        self.ctx._check(lib().exabm4d_comm_max_f64_host(self.ctx.handle, self.handle, ctypes.byref(v)))
        return float(v.value)

    def close(self):
        if self.handle:
            lib().exabm4d_comm_destroy(self.handle)
            self.handle = None


_contexts = {}
_ctx_lock = threading.Lock()
_hip_owner_pid = None        # pid of the process in which this module first created a context
_forked_from_hip_parent = False


def _after_fork_in_child():
    """HIP state does not survive fork(): a child of a process that had already initialised HIP
    (through this module or through torch) cannot use the GPU.  The reference's callers fork
    workers BEFORE any of them touches BM4D (scripts/precompute.py:215-222), which works; the
    other order must fail loudly instead of hanging in the driver."""
    global _forked_from_hip_parent
    torch_mod = __import__("sys").modules.get("torch")
    torch_up = False
    try:
        torch_up = bool(torch_mod is not None and torch_mod.cuda.is_initialized())
    except Exception:
        pass
    if _hip_owner_pid is not None or torch_up:
        _forked_from_hip_parent = True


os.register_at_fork(after_in_child=_after_fork_in_child)


def context(device=None):
    """The calling process's context for ``device`` (default: ``default_device()`` -- EXABM4D_DEVICE,
    LOCAL_RANK, or this pool worker's index mod the device count), created lazily.

    A context inherited through fork() is never reused: HIP state does not survive fork."""
    if device is None:
        device = default_device()
    global _hip_owner_pid
    if _forked_from_hip_parent:
        raise NativeError(
            "this process was fork()ed from a process that had already initialised HIP; the GPU "
            "cannot be used here.  Fork the workers before the parent touches the GPU (as "
            "scripts/precompute.py does), or start them with the 'spawn' method.")
    key = (os.getpid(), int(device))
    with _ctx_lock:
        ctx = _contexts.get(key)
        if ctx is None:
            ctx = Context(device)
            _contexts[key] = ctx
            if _hip_owner_pid is None:
                _hip_owner_pid = os.getpid()
    return ctx


def new_context(device=None):
    """One MORE context of the calling process on ``device`` (its own stream and scratch): for callers that keep
    several device calls in flight from several threads (the broker).  Not cached; the caller keeps it."""
    first = context(device)                 # the fork checks, and the process's cached context
    return Context(first.device)


def check_host_volume_pair(src, dst):
    """What exabm4d_denoise_chunked_u16_host needs of its two host arrays (raises ValueError)."""
    for a in (src, dst):
        if not isinstance(a, np.ndarray) or a.ndim != 3 or a.dtype != np.uint16 or not a.flags.c_contiguous:
            raise ValueError("3-D C-contiguous uint16 arrays expected")
    if src.shape != dst.shape:
        raise ValueError("source and destination differ in shape")
    if not dst.flags.writeable:
        raise ValueError("the destination is read-only")
    if np.shares_memory(src, dst):
        raise ValueError("source and destination may not overlap")


def slab_order_tile(position, columns, q):
    """Tile of launch position ``position`` (a workgroup's ticket, bm_kernels.hip ORDER) in slab order:
    XCD c = position mod 8 walks positions [c q, (c + 1) q) of every slab of ``columns`` tiles; returns
    (slab, column) or None for a padding position.  The host-side mirror of ``xcd_slab_sync``: the tile one
    slab below has position - 8 q, i.e. a smaller ticket."""
    c, j = position & 7, position >> 3
    slab, w = divmod(j, q)
    t = c * q + w
    return (slab, t) if t < columns else None


def blockmatch_plan(shape, batch=1, ctx=None):
    """The launch block matching chooses for this geometry (host logic of libexabm4d.so, no GPU needed):
    dict with the tile grid, the slab-order parameter, whether tiles carry their top cell layer upwards
     and the part of the context's scratch that takes.  ``ctx``: follow that context's
    options (None: the defaults)."""
    out = (ctypes.c_int32 * 6)()
    nbytes = ctypes.c_uint64()
    rc = lib().exabm4d_blockmatch_plan(ctx.handle if ctx is not None else None, int(shape[0]), int(shape[1]),
                                       int(shape[2]), int(batch), out, ctypes.byref(nbytes))
    if rc != 0:
        raise ValueError("blockmatch_plan: %s" % lib().exabm4d_last_error(None).decode())
    return {"tiles_z": out[0], "tiles_y": out[1], "tiles_x": out[2], "slab_order_q": out[3], "carry": bool(out[4]),
            "flat_tiles": bool(out[5]), "carry_bytes": int(nbytes.value)}


def device_count():
    """Number of visible devices.  hipGetDeviceCount initialises the HIP runtime, so the calling
    process counts as a HIP owner for the fork guard from here on."""
    global _hip_owner_pid
    n = int(lib().exabm4d_device_count())
    if _hip_owner_pid is None:
        _hip_owner_pid = os.getpid()
    return n


def device_count_no_init():
    """Number of GPUs this process would see, WITHOUT initialising the HIP runtime (a parent that is about to
    fork workers must not, see ``context``): the visibility lists HIP honours if one is set, else the KFD
    topology (nodes with SIMDs are GPUs).  0 when neither says anything."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return 0
    return n


def worker_index():
    """0-based index of this process among its parent's pool workers (``multiprocessing`` numbers the
    processes it starts: ``current_process()._identity`` -- what ``ProcessPoolExecutor`` workers carry), or
    None for a process nobody numbered (the main process, a plain ``os.fork``)."""
    try:
        import multiprocessing
        ident = multiprocessing.current_process()._identity
        return int(ident[-1]) - 1 if ident else None
    except Exception:
        return None


def default_device(count=None):
    """The device of a caller that named none: ``EXABM4D_DEVICE`` if set; else ``LOCAL_RANK`` (one process
    per GPU under torch.distributed.run); else -- the reference's worker pattern, a pool of forked workers
    that each call ``bm4d()`` (scripts/precompute.py:215-228) -- worker index mod device count, so that a
    pool spreads over the node's GPUs instead of piling onto device 0; a process without a worker index
    (the main process) gets 0.  ``count``: the device count to use (tests); default ``device_count_no_init()``."""
    v = os.environ.get("EXABM4D_DEVICE")
    if v not in (None, ""):
        return int(v)
    v = os.environ.get("LOCAL_RANK")
    if v not in (None, ""):
        return int(v)
    idx = worker_index()
    if idx is None:
        return 0
    n = device_count_no_init() if count is None else int(count)
    return idx % n if n > 0 else 0


# -- host-only helpers (no GPU) -----------------------------------------------------------------
def codec_chunk_bound(n, typesize):
    return int(lib().exabm4d_codec_chunk_bound(int(n), int(typesize)))


def codec_volume_bound(typesize, shape, chunk):
    b = int(lib().exabm4d_codec_volume_bound(int(typesize), int(shape[0]), int(shape[1]),
                                             int(shape[2]), int(chunk[0]), int(chunk[1]),
                                             int(chunk[2])))
    if b == 0:
        raise ValueError("codec: typesize must be 2 or 4 and every size >= 1")
    return b



def grid_positions(n):
    c = lib().exabm4d_grid_count(int(n))
    pos = np.zeros(max(c, 0), dtype=np.int32)
    if c > 0:
        lib().exabm4d_grid_positions(int(n), pos.ctypes.data_as(c_i32p))
    return pos


def tables(params=None):
    p = params or default_params()
    dct = np.zeros(64, dtype=np.float32)
    win = np.zeros(512, dtype=np.float32)
    rc = lib().exabm4d_tables(ctypes.byref(p), dct.ctypes.data_as(c_f32p),
                              win.ctypes.data_as(c_f32p))
    if rc:
        raise ValueError(lib().exabm4d_last_error(None).decode())
    return dct.reshape(8, 8), win.reshape(8, 8, 8)


def match_decode(keys16, ref_pos, ny, nx):
    keys16 = np.ascontiguousarray(keys16, dtype=np.uint32)
    idx = np.zeros(16, dtype=np.int64)
    dist = np.zeros(16, dtype=np.float32)
    cnt = ctypes.c_int()
    rc = lib().exabm4d_match_decode(keys16.ctypes.data_as(c_u32p), int(ref_pos[0]),
                                    int(ref_pos[1]), int(ref_pos[2]), int(ny), int(nx),
                                    idx.ctypes.data_as(c_i64p), dist.ctypes.data_as(c_f32p),
                                    ctypes.byref(cnt))
    if rc:
        raise ValueError(lib().exabm4d_last_error(None).decode())
    return idx, dist, cnt.value
