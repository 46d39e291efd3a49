"""The chunk store's layout on the host (no GPU): an EncodedVolume built from the C restatement of the coder
(oracle/exac_codec.c through oracle/codec_oracle.py) is written as <path>/zarr.json + c/0/0/z/y/x chunk files
-- the key layout of the reference's write_zarr(chunks=(1, 1, 64, 64, 64)) (utils/img_util.py:898-950) -- and
every chunk file decodes, alone, to its (truncated) chunk."""
import json
import os

import numpy as np
import pytest

from aind_exaspim_image_compression.utils import chunk_store as S
from aind_exaspim_image_compression.utils.chunk_codec import EncodedVolume


def _encoded(vol, chunk):
    from oracle import codec_oracle as co
    blobs = [bytes(co.encode(c)) for c in co.chunks(vol, chunk)]
    sizes = np.array([len(b) for b in blobs], dtype=np.uint32)
    offsets = np.zeros(len(blobs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum((sizes.astype(np.uint64) + 15) // 16 * 16)
    data = np.zeros(int(offsets[-1]), dtype=np.uint8)
    for b, o in zip(blobs, offsets[:-1]):
        data[int(o):int(o) + len(b)] = np.frombuffer(b, dtype=np.uint8)
    return EncodedVolume(data, offsets, sizes, vol.shape, chunk, 2), blobs


def test_store_layout_and_chunk_files(tmp_path):
    from oracle import codec_oracle as co
    rng = np.random.default_rng(3)
    vol = np.clip(rng.normal(300, 25, (40, 33, 70)), 0, 65535).astype(np.uint16)     # ragged on all three axes
    chunk = (16, 16, 32)
    enc, blobs = _encoded(vol, chunk)
    path = str(tmp_path / "vol.zarr")
    total = S.write_encoded(enc, path, attributes={"sigma": 24.0})
    assert total == sum(len(b) for b in blobs) == enc.nbytes
    meta = json.load(open(os.path.join(path, "zarr.json")))
    assert meta["zarr_format"] == 3 and meta["node_type"] == "array"
    assert meta["shape"] == [1, 1, 40, 33, 70] and meta["data_type"] == "uint16"
    assert meta["chunk_grid"]["configuration"]["chunk_shape"] == [1, 1, 16, 16, 32]
    assert meta["chunk_key_encoding"] == {"name": "default", "configuration": {"separator": "/"}}
    assert meta["codecs"][0]["name"] == "exac" and meta["codecs"][0]["configuration"]["edge_chunks"] == "truncated"
    assert meta["attributes"]["sigma"] == 24.0
    # one file per chunk under c/0/0/z/y/x, 3 x 3 x 3 of them, each exactly codec.encode(chunk)
    files = sorted(os.path.relpath(os.path.join(d, f), path) for d, _, fs in os.walk(os.path.join(path, "c")) for f in fs)
    assert len(files) == 27 and S.chunk_key(2, 2, 2) in files and S.chunk_key(0, 1, 2) in files
    k = 0
    for iz in range(3):
        for iy in range(3):
            for ix in range(3):
                blob = open(os.path.join(path, S.chunk_key(iz, iy, ix)), "rb").read()
                assert blob == blobs[k]
                want = vol[16 * iz:16 * iz + 16, 16 * iy:16 * iy + 16, 32 * ix:32 * ix + 32]
                np.testing.assert_array_equal(co.decode(blob, want.size, 2)[0].reshape(want.shape), want)   # edge chunks: truncated
                k += 1
    # and back into a container: the same streams at 16-byte aligned offsets
    back, meta2 = S.read_encoded(path)
    assert back.shape == vol.shape and back.chunk == chunk and back.typesize == 2 and meta2 == meta
    assert [back.chunk_bytes(i) for i in range(27)] == blobs and not np.any(back.offsets % 16)


def test_store_reader_rejects_what_it_does_not_understand(tmp_path):
    vol = np.full((8, 8, 8), 7, np.uint16)
    enc, _ = _encoded(vol, (8, 8, 8))
    path = str(tmp_path / "v.zarr")
    S.write_encoded(enc, path)
    meta = json.load(open(os.path.join(path, "zarr.json")))
    for mutate in (lambda m: m.update(zarr_format=2), lambda m: m["codecs"].append({"name": "gzip"}),
                   lambda m: m.update(shape=[2, 1, 8, 8, 8]), lambda m: m.update(data_type="int32"),
                   lambda m: m.pop("chunk_grid")):
        bad = json.loads(json.dumps(meta))
        mutate(bad)
        json.dump(bad, open(os.path.join(path, "zarr.json"), "w"))
        with pytest.raises(ValueError):
            S.read_encoded(path)
    json.dump(meta, open(os.path.join(path, "zarr.json"), "w"))
    os.remove(os.path.join(path, S.chunk_key(0, 0, 0)))
    with pytest.raises(FileNotFoundError):
        S.read_encoded(path)
    with pytest.raises(ValueError):
        S.write_encoded(EncodedVolume(None, enc.offsets, enc.sizes, enc.shape, enc.chunk, 2), path)
