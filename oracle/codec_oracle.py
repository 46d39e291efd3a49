"""ctypes front-end of the chunk entropy coder's CPU restatement (oracle/exac_codec.c).

TEST INFRASTRUCTURE ONLY: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package never does.  The formats it
restates (EXAC v1 and v2) are this repo's specification: the reference's own codec is
third-party Blosc-zstd (reference ``utils/img_util.py:401-441``, ``evaluate.py:40``), absent here,
so parity with its byte counts is unpinned.
"""
import ctypes

import numpy as np

from oracle import bm4d_oracle

_bound = False


def _lib():
    global _bound
    L = bm4d_oracle.lib()
    if not _bound:
        u8p = ctypes.POINTER(ctypes.c_uint8)
        c_sz, c_int = ctypes.c_size_t, ctypes.c_int
        L.orc_exac_bound.argtypes = [c_sz, c_int]
        L.orc_exac_bound.restype = c_sz
        L.orc_exac_normalize.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32,
                                         ctypes.POINTER(ctypes.c_uint16)]
        L.orc_exac_normalize.restype = None
        L.orc_exac_encode.argtypes = [ctypes.c_void_p, c_sz, c_int, u8p]
        L.orc_exac_encode.restype = c_sz
        L.orc_exac_decode.argtypes = [u8p, c_sz, c_sz, c_int, ctypes.c_void_p]
        L.orc_exac_decode.restype = c_sz
        L.orc_exac_check_reciprocal.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        L.orc_exac_check_reciprocal.restype = c_int
        L.orc_exac2_bound.argtypes = [c_sz, c_int]
        L.orc_exac2_bound.restype = c_sz
        L.orc_exac2_normalize.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint16)]
        L.orc_exac2_normalize.restype = None
        L.orc_exac2_encode.argtypes = [ctypes.c_void_p, c_sz, c_sz, c_sz, c_int, u8p]
        L.orc_exac2_encode.restype = c_sz
        L.orc_exac2_decode.argtypes = [u8p, c_sz, c_sz, c_int, ctypes.c_void_p]
        L.orc_exac2_decode.restype = c_sz
        _bound = True
    return L


_DTYPES = {2: np.uint16, 4: np.int32}


def _typesize(a):
    if a.dtype == np.uint16:
        return 2
    if a.dtype == np.int32:
        return 4
    raise ValueError("the chunk coder takes uint16 or int32 elements")


def bound(n, typesize):
    return int(_lib().orc_exac_bound(int(n), int(typesize)))


def normalize(counts):
    """256 symbol counts -> 256 normalised frequencies (sum 4096; all zero for an empty plane)."""
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    f = np.zeros(256, dtype=np.uint16)
    _lib().orc_exac_normalize(c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(c.sum()),
                              f.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    return f


def bound2(n, typesize):
    return int(_lib().orc_exac2_bound(int(n), int(typesize)))


def normalize2(counts):
    """64 symbol counts of one context -> 64 normalised frequencies (EXAC v2 rule)."""
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    assert c.size == 64
    f = np.zeros(64, dtype=np.uint16)
    _lib().orc_exac2_normalize(c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                               f.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    return f


def shape3(shape):
    """The (ez, ey, ex) a chunk of this numpy shape is coded as: leading unit axes dropped, fewer
    than three axes padded in front (same rule as the product's chunk_codec._shape3)."""
    shape = tuple(int(s) for s in shape)
    while len(shape) > 3 and shape[0] == 1:
        shape = shape[1:]
    if len(shape) > 3:
        raise ValueError("expected at most three non-trivial axes")
    return (1,) * (3 - len(shape)) + shape


def encode(chunk, version=2):
    """One chunk (C order) of uint16 / int32 -> its EXAC byte string.  Version 2 (default) models
    the chunk as the 3-D array it is; version 1 (byte planes) only sees the element sequence."""
    a = np.ascontiguousarray(chunk)
    ts = _typesize(a)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    if version == 1:
        out = np.empty(bound(a.size, ts), dtype=np.uint8)
        n = _lib().orc_exac_encode(a.ctypes.data, a.size, ts, out.ctypes.data_as(u8p))
        return out[:n].tobytes()
    if version != 2:
        raise ValueError("EXAC version must be 1 or 2")
    _, ey, ex = shape3(a.shape)
    out = np.empty(bound2(a.size, ts), dtype=np.uint8)
    n = _lib().orc_exac2_encode(a.ctypes.data, a.size, max(ey, 1), max(ex, 1), ts, out.ctypes.data_as(u8p))
    if n == 0:
        raise ValueError("orc_exac2_encode rejected the chunk geometry")
    return out[:n].tobytes()


def decode(data, n, typesize):
    """EXAC byte string (either version, told apart by the header) -> 1-D array of n elements;
    raises ValueError on a malformed stream."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.empty(int(n), dtype=_DTYPES[typesize])
    fn = _lib().orc_exac2_decode if buf.size > 2 and buf[2] == 2 else _lib().orc_exac_decode
    used = fn(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), buf.size, int(n), int(typesize),
              out.ctypes.data)
    if used == 0:
        raise ValueError("malformed EXAC stream")
    return out, int(used)


def check_reciprocal(x, f):
    return bool(_lib().orc_exac_check_reciprocal(int(x), int(f)))


def chunks(vol, chunk):
    """C-order chunk walk of compute_cratio (reference utils/img_util.py:419-438)."""
    vol = np.asarray(vol)
    for z0 in range(0, vol.shape[0], chunk[0]):
        for y0 in range(0, vol.shape[1], chunk[1]):
            for x0 in range(0, vol.shape[2], chunk[2]):
                yield np.ascontiguousarray(vol[z0:z0 + chunk[0], y0:y0 + chunk[1],
                                               x0:x0 + chunk[2]])


def plane_entropy_bytes(chunk):
    """Order-0 entropy bound (bytes) of the byte planes of one chunk: the floor of the coder."""
    a = np.ascontiguousarray(chunk)
    ts = _typesize(a)
    u = a.reshape(-1).astype(np.uint32) if ts == 2 else \
        ((a.reshape(-1).astype(np.int64) << 1) ^ (a.reshape(-1).astype(np.int64) >> 31)).astype(
            np.uint32)
    bits = 0.0
    for p in range(ts):
        h = np.bincount((u >> (8 * p)) & 255, minlength=256).astype(np.float64)
        nz = h[h > 0]
        bits += float(-(nz * np.log2(nz / h.sum())).sum())
    return bits / 8.0
