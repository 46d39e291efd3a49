"""ctypes front-end of the C oracle (oracle/exabm4d_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package never does.  PARITY UNPINNED with respect to the
reference's third-party ``bm4d`` wheel (reference ``machine_learning/data_handling.py:332``);
see the header of the C file and DESIGN.md section 3.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libexabm4d_oracle.so")
_lib = None

DEFAULTS = dict(lambda_ht=2.7, c_match_ht=3.0, c_match_wie=0.6, kaiser_beta=2.0)
KEY_EMPTY = 0xFFFFFFFF


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("exabm4d_oracle.c", "exac_codec.c",
                                             "exabm4d_cpu_port.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(map(os.path.getmtime, srcs)):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        f32p = ctypes.POINTER(ctypes.c_float)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        u16p = ctypes.POINTER(ctypes.c_uint16)
        i32p = ctypes.POINTER(ctypes.c_int32)
        c_int, c_f, c_d, c_sz = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t
        L.orc_grid_count.argtypes = [c_int]
        L.orc_grid_count.restype = c_int
        L.orc_grid_positions.argtypes = [c_int, i32p]
        L.orc_tables.argtypes = [c_d, f32p, f32p]
        L.orc_keymax.argtypes = [c_f, c_f]
        L.orc_keymax.restype = ctypes.c_uint32
        L.orc_blockmatch.argtypes = [f32p, c_int, c_int, c_int, c_f, c_f, u32p]
        L.orc_group_transform.argtypes = [f32p, c_int, c_int]
        L.cpu_group_transform.argtypes = [f32p, c_int, c_int]
        L.cpu_group_transform.restype = None
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.orc_stage.argtypes = [f32p, f32p, u32p, c_int, c_int, c_int, c_f, c_f, c_d, c_int, f32p, f32p]
        L.orc_stage_q.argtypes = [f32p, f32p, u32p, c_int, c_int, c_int, c_f, c_f, c_d, c_int, i64p, i64p]
        L.orc_stage_q.restype = None
        L.orc_data_exp.argtypes = [f32p, c_sz]
        L.orc_data_exp.restype = c_int
        L.orc_rcp_nr.argtypes = [c_f]
        L.orc_rcp_nr.restype = c_f
        L.orc_den_from_corners.argtypes = [i64p, c_int, c_int, c_int, c_d, f32p]
        L.orc_den_from_corners.restype = None
        L.orc_num_to_float.argtypes = [i64p, c_sz, c_int, f32p]
        L.orc_num_to_float.restype = None
        L.orc_normalize.argtypes = [f32p, f32p, f32p, c_sz, c_f, c_f]
        L.orc_bm4d.argtypes = [f32p, f32p, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_d, c_int,
                               c_f, c_f]
        L.orc_bm4d_e.argtypes = L.orc_bm4d.argtypes + [c_int]
        L.orc_bm4d_e.restype = None
        L.orc_bm4d_m.argtypes = L.orc_bm4d.argtypes + [c_int, c_int, c_f]
        L.orc_bm4d_m.restype = None
        L.orc_round_counts.argtypes = [f32p, f32p, c_sz, c_f]
        L.orc_round_counts.restype = None
        L.orc_bm4d_u16.argtypes = [u16p, u16p, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_d,
                                   c_int]
        L.orc_dctq_forward.argtypes = [u16p, c_int, c_int, c_int, c_f, i32p]
        L.orc_dctq_inverse.argtypes = [i32p, c_int, c_int, c_int, c_f, u16p]
        L.orc_dctq_forward.restype = None
        L.orc_dctq_inverse.restype = None
        L.orc_num_threads.restype = c_int
        L.orc_set_threads.argtypes = [c_int]
        L.orc_set_threads.restype = None
        # the CPU baseline port (oracle/exabm4d_cpu_port.c): same signatures as the oracle's
        L.cpu_blockmatch.argtypes = L.orc_blockmatch.argtypes
        L.cpu_stage.argtypes = L.orc_stage.argtypes
        L.cpu_stage_q.argtypes = L.orc_stage_q.argtypes
        L.cpu_stage_q.restype = None
        L.cpu_bm4d.argtypes = L.orc_bm4d.argtypes
        L.cpu_bm4d_u16.argtypes = L.orc_bm4d_u16.argtypes
        for name in ("cpu_blockmatch", "cpu_stage", "cpu_bm4d", "cpu_bm4d_u16"):
            getattr(L, name).restype = None
        for name in ("orc_grid_positions", "orc_tables", "orc_blockmatch", "orc_group_transform",
                     "orc_stage", "orc_normalize", "orc_bm4d", "orc_bm4d_u16"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().orc_num_threads())


def set_threads(n):
    """omp_set_num_threads for the oracle / port library (bench.py's 1-thread and all-core legs)."""
    lib().orc_set_threads(int(n))


def grid_positions(n):
    c = lib().orc_grid_count(int(n))
    pos = np.zeros(c, dtype=np.int32)
    if c:
        lib().orc_grid_positions(int(n), _p(pos, ctypes.c_int32))
    return pos


def tables(beta=DEFAULTS["kaiser_beta"]):
    dct = np.zeros(64, dtype=np.float32)
    win = np.zeros(512, dtype=np.float32)
    lib().orc_tables(float(beta), _p(dct, ctypes.c_float), _p(win, ctypes.c_float))
    return dct.reshape(8, 8), win.reshape(8, 8, 8)


def keymax(sigma, c_match):
    return int(lib().orc_keymax(float(sigma), float(c_match)))


def blockmatch(vol, sigma, c_match=DEFAULTS["c_match_ht"], port=False):
    """-> keys [gz,gy,gx,16] uint32.  ``port``: the CPU baseline port instead of
    the oracle (bit-identical tables)."""
    vol = _f32(vol)
    nz, ny, nx = vol.shape
    g = [len(grid_positions(n)) for n in (nz, ny, nx)]
    keys = np.empty((g[0], g[1], g[2], 16), dtype=np.uint32)
    fn = lib().cpu_blockmatch if port else lib().orc_blockmatch
    fn(_p(vol, ctypes.c_float), nz, ny, nx, float(sigma), float(c_match), _p(keys, ctypes.c_uint32))
    return keys


def group_transform(g, inverse=False, port=False):
    """In-place-semantics 4-D transform of a [K,8,8,8] group; returns a new array."""
    g = _f32(g).copy()
    fn = lib().cpu_group_transform if port else lib().orc_group_transform
    fn(_p(g, ctypes.c_float), int(g.shape[0]), int(bool(inverse)))
    return g


AUTO_EXP = -2 ** 31      # data_exp: take E from the noisy volume (the fp32 entry points' rule)
U16_DATA_EXP = 17        # E of the uint16 entry points
MAX_DATA_EXP = 56        # fp32 entry points: beyond |v| < 2^56 squares of coefficients leave fp32


def data_exp(vol):
    """E with max |v| < 2^E, from the largest |v| bit pattern."""
    vol = _f32(vol)
    return int(lib().orc_data_exp(_p(vol, ctypes.c_float), vol.size))


def rcp_nr(d):
    """R(d) of DESIGN.md 3.7 for one fp32 value."""
    return float(lib().orc_rcp_nr(float(d)))


def stage(noisy, keys, sigma, basic=None, lambda_ht=DEFAULTS["lambda_ht"],
          beta=DEFAULTS["kaiser_beta"], port=False, data_exp=None):
    """-> (num, den) of one collaborative-filtering stage (hard-threshold if basic is None):
    num = fl32(NUM 2^(E-43)), den = corner weights (*) window.  ``data_exp`` None:
    E from ``noisy``."""
    noisy = _f32(noisy)
    nz, ny, nx = noisy.shape
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    num = np.zeros_like(noisy)
    den = np.zeros_like(noisy)
    bp = None
    if basic is not None:
        basic = _f32(basic)
        bp = _p(basic, ctypes.c_float)
    fn = lib().cpu_stage if port else lib().orc_stage
    fn(_p(noisy, ctypes.c_float), bp, _p(keys, ctypes.c_uint32), nz, ny, nx, float(sigma),
       float(lambda_ht), float(beta), AUTO_EXP if data_exp is None else int(data_exp),
       _p(num, ctypes.c_float), _p(den, ctypes.c_float))
    return num, den


def stage_q(noisy, keys, sigma, data_exp, basic=None, lambda_ht=DEFAULTS["lambda_ht"],
            beta=DEFAULTS["kaiser_beta"], port=False):
    """-> (NUM, CW): the integer sums of DESIGN.md 3.8 (numerator in units of 2^(E-43), group
    weights on block corners in units of 2^-40)."""
    noisy = _f32(noisy)
    nz, ny, nx = noisy.shape
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    NUM = np.zeros(noisy.shape, dtype=np.int64)
    CW = np.zeros(noisy.shape, dtype=np.int64)
    bp = None
    if basic is not None:
        basic = _f32(basic)
        bp = _p(basic, ctypes.c_float)
    fn = lib().cpu_stage_q if port else lib().orc_stage_q
    fn(_p(noisy, ctypes.c_float), bp, _p(keys, ctypes.c_uint32), nz, ny, nx, float(sigma),
       float(lambda_ht), float(beta), int(data_exp), _p(NUM, ctypes.c_int64), _p(CW, ctypes.c_int64))
    return NUM, CW


def normalize(num, den, clip=None):
    num, den = _f32(num), _f32(den)
    out = np.empty_like(num)
    lo, hi = (1.0, 0.0) if clip is None else clip
    lib().orc_normalize(_p(num, ctypes.c_float), _p(den, ctypes.c_float), _p(out, ctypes.c_float),
                        num.size, float(lo), float(hi))
    return out


def round_counts(vol, offset):
    """What stage 2 of the uint16 form matches on: fl(rint(clamp(v + offset, 0, 65535))) -
    offset, element by element."""
    vol = _f32(vol)
    out = np.empty_like(vol)
    lib().orc_round_counts(_p(vol, ctypes.c_float), _p(out, ctypes.c_float), vol.size, float(offset))
    return out


def bm4d(vol, sigma, stages=2, clip=None, data_exp=None, match_counts_offset=None, **kw):
    """Whole two-stage pipeline on one fp32 volume.  ``data_exp``: E of DESIGN.md 3.8 (None: from the
    volume, as the fp32 entry points do; 17 reproduces the uint16 entry points).  ``match_counts_offset``:
    stage 2 matches on ``round_counts(basic, offset)`` as the uint16 entry points do (None: on the basic
    estimate itself, the fp32 entry points)."""
    p = {**DEFAULTS, **kw}
    vol = _f32(vol)
    nz, ny, nx = vol.shape
    if data_exp is None and int(lib().orc_data_exp(_p(vol, ctypes.c_float), vol.size)) > MAX_DATA_EXP:
        raise ValueError("fp32 volume outside the working range of DESIGN.md 3.8 (|v| >= 2^56, inf or NaN)")
    out = np.empty_like(vol)
    lo, hi = (1.0, 0.0) if clip is None else clip
    lib().orc_bm4d_m(_p(vol, ctypes.c_float), _p(out, ctypes.c_float), nz, ny, nx, float(sigma),
                     float(p["lambda_ht"]), float(p["c_match_ht"]), float(p["c_match_wie"]),
                     float(p["kaiser_beta"]), int(stages), float(lo), float(hi),
                     AUTO_EXP if data_exp is None else int(data_exp),
                     0 if match_counts_offset is None else 1,
                     0.0 if match_counts_offset is None else float(match_counts_offset))
    return out


def bm4d_u16(vol, sigma, offset, stages=2, port=False, **kw):
    p = {**DEFAULTS, **kw}
    vol = np.ascontiguousarray(vol, dtype=np.uint16)
    nz, ny, nx = vol.shape
    out = np.empty_like(vol)
    (lib().cpu_bm4d_u16 if port else lib().orc_bm4d_u16)(_p(vol, ctypes.c_uint16), _p(out, ctypes.c_uint16), nz, ny, nx,
                       float(sigma), float(offset), float(p["lambda_ht"]), float(p["c_match_ht"]),
                       float(p["c_match_wie"]), float(p["kaiser_beta"]), int(stages))
    return out


def dctq_forward(vol_u16, q):
    """DESIGN.md 3.10 transform quantiser -> int32 indices [nbz, nby, nbx, 512]."""
    v = np.ascontiguousarray(vol_u16, dtype=np.uint16)
    nz, ny, nx = v.shape
    nb = [-(-n // 8) for n in v.shape]
    out = np.empty((nb[0], nb[1], nb[2], 512), dtype=np.int32)
    lib().orc_dctq_forward(_p(v, ctypes.c_uint16), nz, ny, nx, float(q), _p(out, ctypes.c_int32))
    return out


def dctq_inverse(idx, shape, q):
    i = np.ascontiguousarray(idx, dtype=np.int32)
    nz, ny, nx = shape
    out = np.empty(shape, dtype=np.uint16)
    lib().orc_dctq_inverse(_p(i, ctypes.c_int32), nz, ny, nx, float(q), _p(out, ctypes.c_uint16))
    return out


def decode_keys(keys16):
    """-> list of (dz,dy,dx), quantised distance S/512, for the valid entries."""
    out = []
    for k in np.asarray(keys16, dtype=np.uint32):
        if int(k) == KEY_EMPTY:
            break
        code = int(k) & 0x7FF
        if code == 0:
            d = (0, 0, 0)
        else:
            l = code - 1
            d = (l // 121 - 5, (l // 11) % 11 - 5, l % 11 - 5)
        s = np.array([int(k) & 0xFFFFF800], dtype=np.uint32).view(np.float32)[0]
        out.append((d, float(s) / 512.0))
    return out


def padded_chunks(vol, chunk, halo, core=None):
    """Chunk-local mode: yields ``(slices of the core, halo in
    front per axis, padded array)`` for every chunk of the core planes ``core = (zc0, zc1)`` of
    ``vol`` -- cores of ``chunk`` voxels (ragged last ones), read with ``halo`` voxels per side,
    the read window clamped to the array (no padding is invented at the faces)."""
    vol = np.asarray(vol)
    zc0, zc1 = (0, vol.shape[0]) if core is None else core
    lo = (zc0, 0, 0)
    hi = (zc1, vol.shape[1], vol.shape[2])
    for z0 in range(lo[0], hi[0], chunk):
        for y0 in range(lo[1], hi[1], chunk):
            for x0 in range(lo[2], hi[2], chunk):
                o = (z0, y0, x0)
                e = [min(chunk, h - s) for s, h in zip(o, hi)]
                front = [min(halo, s) for s in o]
                win = tuple(slice(s - f, min(dim, s + n + halo))
                            for s, n, f, dim in zip(o, e, front, vol.shape))
                yield (tuple(slice(s, s + n) for s, n in zip(o, e)), front,
                       np.ascontiguousarray(vol[win]))


def bm4d_u16_chunked(vol, sigma, offset, chunk, halo, stages=2, core=None, **kw):
    """The identical padded arrays the chunk-local device call processes, one oracle pipeline
    each; only the cores are written.  Returns the core planes."""
    vol = np.ascontiguousarray(vol, dtype=np.uint16)
    zc0, zc1 = (0, vol.shape[0]) if core is None else core
    out = np.zeros((zc1 - zc0,) + vol.shape[1:], dtype=np.uint16)
    for sl, front, padded in padded_chunks(vol, chunk, halo, core):
        den = bm4d_u16(padded, sigma, offset, stages=stages, **kw)
        inner = tuple(slice(f, f + (s.stop - s.start)) for f, s in zip(front, sl))
        out[(slice(sl[0].start - zc0, sl[0].stop - zc0),) + sl[1:]] = den[inner]
    return out
