"""Byte shuffle + zstd through the system's libzstd (ctypes): the codec FAMILY the reference ships
for its compression ratios -- numcodecs.blosc.Blosc(cname="zstd", clevel=5, shuffle=SHUFFLE)
(reference evaluate.py:40, scripts/evaluate_bm4dnet.py:140; clevel 6 at train.py:105).

TEST INFRASTRUCTURE ONLY (tests/, bench.py's out-of-timed-region comparison legs and
cpu_baseline): the product never imports this module.  numcodecs / c-blosc are absent from this
image; libzstd.so.1 (1.4.8, runtime only, no headers) is present, so the comparison is
SHUFFLE + one zstd frame per chunk -- Blosc additionally splits a chunk into blocks and adds a
16-byte header, both of which make its output slightly larger than what is measured here, never
smaller.  `available()` is False where the library is missing and callers skip.
"""
import ctypes
import ctypes.util

import numpy as np

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    for name in ("libzstd.so.1", ctypes.util.find_library("zstd")):
        if not name:
            continue
        try:
            L = ctypes.CDLL(name)
            L.ZSTD_compressBound.restype = ctypes.c_size_t
            L.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
            L.ZSTD_compress.restype = ctypes.c_size_t
            L.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                        ctypes.c_size_t, ctypes.c_int]
            L.ZSTD_decompress.restype = ctypes.c_size_t
            L.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                          ctypes.c_size_t]
            L.ZSTD_isError.restype = ctypes.c_uint
            L.ZSTD_isError.argtypes = [ctypes.c_size_t]
            L.ZSTD_versionNumber.restype = ctypes.c_uint
            _lib = L
            break
        except OSError:
            continue
    return _lib


def available():
    return _load() is not None


def version():
    v = _load().ZSTD_versionNumber()
    return "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100)


def shuffle(chunk):
    """Blosc's SHUFFLE filter: byte p of every element, plane after plane."""
    a = np.ascontiguousarray(chunk)
    return np.ascontiguousarray(a.reshape(-1).view(np.uint8).reshape(-1, a.itemsize).T).reshape(-1)


def compress(buf, level=5):
    L = _load()
    src = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    cap = L.ZSTD_compressBound(src.size)
    dst = np.empty(cap, dtype=np.uint8)
    n = L.ZSTD_compress(dst.ctypes.data, cap, src.ctypes.data, src.size, int(level))
    if L.ZSTD_isError(n):
        raise RuntimeError("ZSTD_compress failed")
    return dst[:n]


def decompress(buf, nbytes):
    L = _load()
    src = np.ascontiguousarray(buf, dtype=np.uint8)
    dst = np.empty(int(nbytes), dtype=np.uint8)
    n = L.ZSTD_decompress(dst.ctypes.data, dst.size, src.ctypes.data, src.size)
    if L.ZSTD_isError(n) or n != dst.size:
        raise RuntimeError("ZSTD_decompress failed")
    return dst


def shuffle_zstd_size(chunk, level=5):
    """len(Blosc(zstd, level, SHUFFLE).encode(chunk)) up to Blosc's own framing (see above)."""
    return int(compress(shuffle(chunk), level).size)


def volume_size(vol, chunk=(64, 64, 64), level=5, threads=1):
    """Sum of shuffle_zstd_size over the C-order chunk walk of compute_cratio
    (reference utils/img_util.py:419-438); `threads` > 1 codes chunks concurrently (zstd releases
    the GIL inside ctypes calls)."""
    vol = np.asarray(vol)
    pieces = [(z, y, x) for z in range(0, vol.shape[0], chunk[0]) for y in range(0, vol.shape[1], chunk[1])
              for x in range(0, vol.shape[2], chunk[2])]

    def one(p):
        z, y, x = p
        return shuffle_zstd_size(vol[z:z + chunk[0], y:y + chunk[1], x:x + chunk[2]], level)

    if threads <= 1:
        return sum(one(p) for p in pieces)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as ex:
        return sum(ex.map(one, pieces))
