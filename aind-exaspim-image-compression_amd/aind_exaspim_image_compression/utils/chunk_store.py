"""A local chunk store for the coded volume, in the layout the reference writes its results in.

The reference ends its production path with ``write_zarr(img, output_path, chunks=(1, 1, 64, 64, 64), ...)``
(reference utils/img_util.py:898-950; called from scripts/evaluate_bm4dnet.py): a Zarr v3 array, 5-D
(t, c, z, y, x), one object per 64^3 chunk under ``c/0/0/<z>/<y>/<x>``, each compressed with
Blosc(zstd, shuffle).  Here the chunk streams are EXAC (``utils/chunk_codec.py``), coded on
the MI355X in one batched call, and they go to disk in the same directory layout:

    <path>/zarr.json            array metadata (Zarr v3 keys: shape, data_type, chunk_grid, chunk_key_encoding,
                                fill_value, codecs = [{"name": "exac", "configuration": {...}}])
    <path>/c/0/0/<z>/<y>/<x>    the EXAC stream of chunk (z, y, x): exactly ``codec.encode(chunk)``

so that a chunk is addressable by its key like any Zarr chunk and ``sum(file sizes)`` is the denominator of
``compute_cratio`` (utils/img_util.py:401-441).  Differences from a stock Zarr array, stated in the metadata:
the codec is this repo's (a generic Zarr reader needs an ``exac`` codec plug-in: ``ExacCodec.decode``), and
edge chunks are stored TRUNCATED to the array (the reference's ``compute_cratio`` slices them that way; an
EXAC stream carries its own chunk shape), not padded to the chunk grid.  Local file system only -- cloud
stores (fsspec / s3 / gs, which ``write_zarr`` accepts) are out of scope.
"""
import json
import os

import numpy as np

from aind_exaspim_image_compression.utils.chunk_codec import EncodedVolume, ExacCodec

FORMAT_NOTE = ("EXAC chunk streams (aind-exaspim-image-compression_amd, DESIGN.md 3.11b); edge chunks truncated to "
               "the array; decode with utils.chunk_codec.ExacCodec.decode or utils.chunk_store.read_zarr")


def _grid(shape3, chunk3):
    return tuple(-(-s // c) for s, c in zip(shape3, chunk3))


def chunk_key(iz, iy, ix):
    """Zarr v3 default chunk key encoding ("/" separator) of chunk (0, 0, iz, iy, ix) of a 5-D array."""
    return os.path.join("c", "0", "0", str(int(iz)), str(int(iy)), str(int(ix)))


def metadata(shape3, chunk3, typesize=2, version=2, attributes=None):
    """The ``zarr.json`` document of a stored volume."""
    dtype = {2: "uint16", 4: "int32"}[int(typesize)]
    return {
        "zarr_format": 3,
        "node_type": "array",
        "shape": [1, 1] + [int(s) for s in shape3],
        "data_type": dtype,
        "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [1, 1] + [int(c) for c in chunk3]}},
        "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}},
        "fill_value": 0,
        "codecs": [{"name": "exac", "configuration": {"version": int(version), "typesize": int(typesize),
                                                      "edge_chunks": "truncated"}}],
        "attributes": dict(attributes or {}, exac_note=FORMAT_NOTE),
        "dimension_names": ["t", "c", "z", "y", "x"],
    }


def write_encoded(enc, output_path, version=2, attributes=None, overwrite=True):
    """``EncodedVolume`` (host container: data, offsets, sizes) -> chunk store at ``output_path``.  Host work
    only.  Returns the number of bytes written as chunk streams."""
    if enc.data is None:
        raise ValueError("the EncodedVolume carries sizes only (encode with want_bytes=True)")
    gz, gy, gx = _grid(enc.shape, enc.chunk)
    if len(enc.sizes) != gz * gy * gx:
        raise ValueError("EncodedVolume: chunk count does not match its shape and chunk")
    if os.path.exists(os.path.join(output_path, "zarr.json")) and not overwrite:
        raise FileExistsError(output_path)
    os.makedirs(output_path, exist_ok=True)
    total = 0
    k = 0
    for iz in range(gz):
        for iy in range(gy):
            d = os.path.join(output_path, "c", "0", "0", str(iz), str(iy))
            os.makedirs(d, exist_ok=True)
            for ix in range(gx):
                blob = enc.chunk_bytes(k)
                with open(os.path.join(d, str(ix)), "wb") as f:
                    f.write(blob)
                total += len(blob)
                k += 1
    with open(os.path.join(output_path, "zarr.json"), "w") as f:      # last: a store without it is incomplete
        json.dump(metadata(enc.shape, enc.chunk, enc.typesize, version, attributes), f, indent=1)
    return total


def read_encoded(path):
    """Chunk store -> ``(EncodedVolume, metadata)``: the chunk files gathered into one container (every stream
    at a multiple of 16 bytes, as the device decoder takes it).  Host work only; raises ``ValueError`` for
    a store that is not an EXAC array of this layout and ``FileNotFoundError`` for a missing chunk."""
    with open(os.path.join(path, "zarr.json")) as f:
        meta = json.load(f)
    try:
        if meta["zarr_format"] != 3 or meta["node_type"] != "array":
            raise ValueError("not a Zarr v3 array")
        codecs = meta["codecs"]
        if len(codecs) != 1 or codecs[0]["name"] != "exac":
            raise ValueError("the array's codec chain is not [exac]")
        cfg = codecs[0]["configuration"]
        typesize = int(cfg["typesize"])
        shape5, chunk5 = meta["shape"], meta["chunk_grid"]["configuration"]["chunk_shape"]
        if len(shape5) != 5 or shape5[:2] != [1, 1] or len(chunk5) != 5 or chunk5[:2] != [1, 1]:
            raise ValueError("only (1, 1, z, y, x) arrays with (1, 1, cz, cy, cx) chunks are stored this way")
        if meta["data_type"] != {2: "uint16", 4: "int32"}[typesize]:
            raise ValueError("data_type does not match the codec's typesize")
        if meta["chunk_key_encoding"]["name"] != "default" or \
                meta["chunk_key_encoding"]["configuration"].get("separator", "/") != "/":
            raise ValueError("unsupported chunk key encoding")
    except (KeyError, TypeError) as e:
        raise ValueError(f"zarr.json lacks a field this reader needs: {e}") from None
    shape3, chunk3 = tuple(int(s) for s in shape5[2:]), tuple(int(c) for c in chunk5[2:])
    if min(shape3) < 1 or min(chunk3) < 1:
        raise ValueError("empty array or chunk")
    gz, gy, gx = _grid(shape3, chunk3)
    blobs = []
    for iz in range(gz):
        for iy in range(gy):
            for ix in range(gx):
                with open(os.path.join(path, chunk_key(iz, iy, ix)), "rb") as f:
                    blobs.append(f.read())
    sizes = np.array([len(b) for b in blobs], dtype=np.uint32)
    offsets = np.zeros(len(blobs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum((sizes.astype(np.uint64) + 15) // 16 * 16)
    data = np.zeros(int(offsets[-1]), dtype=np.uint8)
    for b, o in zip(blobs, offsets[:-1]):
        data[int(o):int(o) + len(b)] = np.frombuffer(b, dtype=np.uint8)
    chunk_eff = tuple(min(c, s) for c, s in zip(chunk3, shape3))
    return EncodedVolume(data, offsets, sizes, shape3, chunk_eff, typesize), meta


def write_zarr(img, output_path, chunks=(1, 1, 64, 64, 64), codec=None, attributes=None):
    """The reference's ``write_zarr(img, output_path, chunks=(1, 1, 64, 64, 64))`` with the EXAC chunk coder in
    the place of Blosc (utils/img_util.py:898-950): ``img`` (uint16, promoted to 5-D like the reference; t and
    c must be 1) is coded on the device in one batched call and written as a chunk store.  Returns the
    compression ratio raw bytes / stored chunk bytes -- what ``compute_cratio(img, codec, chunks[2:])`` reports,
    unrounded."""
    img = np.asarray(img)
    while img.ndim < 5:
        img = img[np.newaxis, ...]
    if img.ndim != 5 or img.shape[0] != 1 or img.shape[1] != 1:
        raise ValueError("write_zarr stores (1, 1, z, y, x) arrays")
    if len(chunks) != 5 or tuple(chunks[:2]) != (1, 1):
        raise ValueError("chunks must be (1, 1, cz, cy, cx)")
    codec = codec or ExacCodec(2)
    vol = np.ascontiguousarray(img[0, 0])
    enc = codec.encode_volume(vol, chunk=tuple(int(c) for c in chunks[2:]))
    stored = write_encoded(enc, output_path, version=codec.version, attributes=attributes)
    return vol.nbytes / stored


def read_zarr(path, codec=None):
    """Chunk store -> the (1, 1, z, y, x) array, decoded on the device in one call."""
    enc, meta = read_encoded(path)
    codec = codec or ExacCodec(enc.typesize)
    return codec.decode_volume(enc)[np.newaxis, np.newaxis]


def read_chunk(path, iz, iy, ix, codec=None):
    """One chunk by its key, decoded alone (random access: what a viewer does)."""
    with open(os.path.join(path, "zarr.json")) as f:
        meta = json.load(f)
    cfg = meta["codecs"][0]["configuration"]
    shape3 = meta["shape"][2:]
    chunk3 = meta["chunk_grid"]["configuration"]["chunk_shape"][2:]
    ext = tuple(min(c, s - i * c) for i, c, s in zip((iz, iy, ix), chunk3, shape3))
    if min(ext) < 1:
        raise IndexError("chunk index outside the array")
    with open(os.path.join(path, chunk_key(iz, iy, ix)), "rb") as f:
        blob = f.read()
    codec = codec or ExacCodec(int(cfg["typesize"]))
    return codec.decode(blob).reshape(ext)
