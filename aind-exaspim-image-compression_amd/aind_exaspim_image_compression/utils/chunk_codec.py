"""Chunk entropy coder of the encode half (BASELINE config 5).

The reference measures its compression ratio by handing every C-order 64^3 chunk of the uint16
volume to a codec object, ``len(codec.encode(chunk))`` (reference ``utils/img_util.py:401-441``);
the object it passes is ``numcodecs.blosc.Blosc(cname="zstd", clevel=5|6, shuffle=SHUFFLE)``
(``evaluate.py:40``, ``train.py:105``, ``scripts/evaluate_bm4dnet.py:140``) and the same codec
compresses the chunks ``write_zarr`` stores (``utils/img_util.py:935-950``).  ``ExacCodec`` is an
object of that shape -- ``encode(buf) -> bytes``, ``decode(bytes) -> ndarray`` -- whose arithmetic
runs on the MI355X.  zstd itself is third-party and absent, so the byte counts are this codec's:

* EXAC v2 (default): prediction from the voxel above and the voxel in the plane
  before, 64-symbol residual alphabet + raw mantissa bits, 16 static context tables per chunk,
  64 interleaved rANS states.  Smaller than byte shuffle + zstd-5 on the volumes this path
  produces (``tests/test_codec_vs_zstd.py``).
* EXAC v1 (``version=1``): Blosc's byte shuffle followed by a static order-0 rANS
  coder per byte plane -- round 2's format, still written on request and always decoded.

Besides the per-chunk calls the codec codes all chunks of a volume that is already in HBM with one
kernel sequence (``encode_volume``), which is what ``compute_cratio`` uses when it is given this
codec.
"""
import numpy as np

from aind_exaspim_image_compression import _native

_DTYPES = {2: np.dtype(np.uint16), 4: np.dtype(np.int32)}


def _typesize(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.uint16:
        return 2
    if dtype == np.int32:
        return 4
    raise ValueError("ExacCodec codes uint16 or int32 elements, not %s" % dtype)


def _shape3(shape):
    """(ez, ey, ex) of an array shape: leading unit axes dropped (the reference's chunks are
    (1, 1, 64, 64, 64) in ``write_zarr``), fewer than three axes padded in front."""
    shape = tuple(int(s) for s in shape)
    if len(shape) > 3:
        lead = int(np.prod(shape[:-3]))
        if lead != 1:
            raise ValueError("expected at most three non-trivial axes")
        shape = shape[-3:]
    return (1,) * (3 - len(shape)) + shape


class EncodedVolume:
    """All chunk streams of one volume: ``data`` (uint8 container, every stream starting at a
    multiple of 16 bytes), ``offsets`` (uint64, nchunks + 1), ``sizes`` (uint32, exact stream
    lengths = ``len(codec.encode(chunk))``), chunks in (z, y, x) raster order."""

    def __init__(self, data, offsets, sizes, shape, chunk, typesize):
        self.data = data
        self.offsets = offsets
        self.sizes = sizes
        self.shape = tuple(shape)
        self.chunk = tuple(chunk)
        self.typesize = int(typesize)

    @property
    def nbytes(self):
        """Sum of the chunk stream lengths (the denominator of the compression ratio)."""
        return int(self.sizes.sum(dtype=np.uint64))

    def chunk_bytes(self, i):
        o = int(self.offsets[i])
        return self.data[o:o + int(self.sizes[i])].tobytes()


class ExacCodec:
    """numcodecs-shaped codec whose arithmetic runs on the GPU (EXAC v2 by default)."""

    codec_id = "exac"

    def __init__(self, typesize=2, device=None, version=2):
        if typesize not in (2, 4):
            raise ValueError("typesize must be 2 (uint16) or 4 (int32)")
        if version not in (1, 2):
            raise ValueError("EXAC version must be 1 or 2")
        self.typesize = int(typesize)
        self.device = device
        self.version = int(version)

    def get_config(self):
        return {"id": self.codec_id, "typesize": self.typesize, "version": self.version}

    # -- one chunk ---------------------------------------------------------------------------
    def encode(self, buf):
        """One chunk -> bytes.  Version 2 models the chunk as the (up to) 3-D array it is -- pass
        the chunk with its shape, as ``compute_cratio`` does; version 1 only sees the C-order
        element sequence."""
        a = np.ascontiguousarray(buf)
        if _typesize(a.dtype) != self.typesize:
            raise ValueError("element type does not match the codec's typesize")
        if a.size == 0:
            raise ValueError("cannot encode an empty chunk")
        shape = _shape3(a.shape) if self.version == 2 else (1, 1, a.size)
        enc = self.encode_volume(a.reshape(shape), chunk=shape)
        return enc.chunk_bytes(0)

    def decode(self, buf, out=None):
        """bytes of one chunk (either version) -> 1-D array (or filled ``out``) of the codec's
        element type."""
        raw = np.frombuffer(bytes(buf), dtype=np.uint8)
        if raw.size < 8 or raw[0] != ord("E") or raw[1] != ord("X") or raw[3] != self.typesize:
            raise ValueError("not an EXAC stream of this codec's typesize")
        n = int(raw[4:8].view("<u4")[0])
        if n == 0 or n > (1 << 28):
            raise ValueError("EXAC stream with an implausible element count")
        shape = (1, 1, n)
        if raw[2] == 2:
            if raw.size < 16:
                raise ValueError("truncated EXAC v2 header")
            ey, ex = (int(v) for v in raw[8:16].view("<u4"))
            if ey < 1 or ex < 1 or n % (ey * ex):
                raise ValueError("EXAC v2 header with an inconsistent chunk shape")
            shape = (n // (ey * ex), ey, ex)
        pad = (-raw.size) % 16
        data = np.concatenate([raw, np.zeros(pad, np.uint8)]) if pad else raw
        enc = EncodedVolume(data, np.array([0, raw.size], dtype=np.uint64),
                            np.array([raw.size], dtype=np.uint32), shape, shape, self.typesize)
        res = self.decode_volume(enc).reshape(-1)
        if out is not None:
            np.copyto(np.asarray(out).reshape(-1), res)
            return out
        return res

    # -- a whole volume ------------------------------------------------------------------------
    def encode_volume(self, vol, chunk=(64, 64, 64), want_bytes=True):
        """Host array (uint16 / int32, up to 3-D) -> ``EncodedVolume`` on the host.  With
        ``want_bytes=False`` only the sizes are produced (``data`` is None)."""
        a = np.ascontiguousarray(vol)
        if _typesize(a.dtype) != self.typesize:
            raise ValueError("element type does not match the codec's typesize")
        shape = _shape3(a.shape)
        ctx = _native.context(self.device)
        d_vol = ctx.to_device(a.reshape(-1))
        try:
            return self.encode_device(ctx, d_vol, shape, chunk, want_bytes)
        finally:
            d_vol.free()

    def encode_device(self, ctx, d_vol, shape, chunk=(64, 64, 64), want_bytes=True):
        """The same for a volume that already lies in HBM (``d_vol``: device pointer holder)."""
        shape = _shape3(shape)
        chunk = tuple(min(int(c), s) for c, s in zip(_shape3(chunk), shape))
        nchunks = int(np.prod([-(-s // c) for s, c in zip(shape, chunk)]))
        d_sizes = ctx.alloc(4 * nchunks)
        d_off = ctx.alloc(8 * (nchunks + 1))
        d_out = None
        try:
            cap = 0
            if want_bytes:
                cap = _native.codec_volume_bound(self.typesize, shape, chunk)
                d_out = ctx.alloc(cap)
            # the format travels with the call (round 4): no shared context state is touched
            _, container = ctx.codec_encode(d_vol, self.typesize, shape, chunk, out=d_out, out_capacity=cap,
                                            offsets=d_off, sizes=d_sizes, version=self.version)
            sizes = d_sizes.download((nchunks,), np.uint32)
            offsets = d_off.download((nchunks + 1,), np.uint64)
            data = d_out.download((container,), np.uint8) if want_bytes else None
        finally:
            d_sizes.free()
            d_off.free()
            if d_out is not None:
                d_out.free()
        return EncodedVolume(data, offsets, sizes, shape, chunk, self.typesize)

    def decode_volume(self, enc):
        """``EncodedVolume`` -> host array of ``enc.shape``."""
        if enc.typesize != self.typesize:
            raise ValueError("EncodedVolume of another typesize")
        offsets = np.ascontiguousarray(enc.offsets, dtype=np.uint64)
        data = np.ascontiguousarray(enc.data, dtype=np.uint8)
        nchunks = int(np.prod([-(-s // c) for s, c in zip(enc.shape, enc.chunk)]))
        # validate the container on the host as well: a corrupt one must raise, not fault
        if offsets.size != nchunks + 1 or np.any(np.diff(offsets.astype(np.int64)) < 0) or \
                int(offsets[-1]) > data.size or np.any(offsets % 2):
            raise ValueError("malformed EXAC container: offsets are not ascending inside the data")
        n = int(np.prod(enc.shape))
        if n < 1 or n > (1 << 40):
            raise ValueError("malformed EXAC container: implausible volume shape")
        ctx = _native.context(self.device)
        d_in = ctx.to_device(data)
        d_off = ctx.to_device(offsets)
        d_vol = ctx.alloc(n * self.typesize)
        try:
            ctx.codec_decode(d_in, data.size, d_off, self.typesize, enc.shape, enc.chunk, d_vol)
            return d_vol.download(enc.shape, _DTYPES[self.typesize])
        finally:
            d_in.free()
            d_off.free()
            d_vol.free()

    def chunk_sizes(self, vol, chunk=(64, 64, 64)):
        """``len(self.encode(c))`` of every chunk of ``vol``, one batched device call."""
        return self.encode_volume(vol, chunk, want_bytes=False).sizes


class ShuffleRansCodec(ExacCodec):
    """Round 2's name of the codec, kept for callers that pinned it: the byte-shuffle + per-plane
    order-0 rANS format (EXAC v1).  New code uses ``ExacCodec`` (v2)."""

    codec_id = "exac-shuffle-rans"

    def __init__(self, typesize=2, device=None):
        super().__init__(typesize, device, version=1)
