"""The RCCL halo exchange behind the C-ABI (csrc/comm_rccl.hip) and the torch-free
sharded paths above it (distributed.*_native).  One MI355X: a one-rank communicator exercises the whole
plumbing -- dlopen of librccl, ncclGetUniqueId, ncclCommInitRank, grouped ncclSend / ncclRecv to oneself on
the context's stream, the all-reduce barrier -- and the slab / chunk drivers with world = 1; the two-rank
cases need two devices (RCCL admits one rank per device) and are skipped here."""
import os
import subprocess
import sys

import numpy as np
import pytest

from util import synth_volume

from aind_exaspim_image_compression import _native
from aind_exaspim_image_compression import distributed as D

pytestmark = pytest.mark.gpu
SIGMA, OFFSET = 24.0, 37.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm(ctx):
    c = _native.Comm(ctx, 1, 0, _native.comm_unique_id())
    yield c
    c.close()


def test_grouped_send_recv_on_the_contexts_stream(ctx, comm):
    """Peers = oneself: the first send of the group meets the first receive, the second the second; the
    exchange is ordered on the context's stream between the fill before it and the copy after it."""
    rng = np.random.default_rng(0)
    a = rng.integers(0, 65536, 5 * 40 * 44).astype(np.uint16)          # five uint16 planes
    b = rng.normal(0, 1, 3 * 40 * 44).astype(np.float32)               # three fp32 planes, as bytes
    d_a, d_b = ctx.to_device(a), ctx.to_device(b)
    r_a, r_b = ctx.alloc(a.nbytes).fill(0xFF), ctx.alloc(b.nbytes).fill(0xFF)
    comm.halo_exchange(0, d_a, r_a, a.nbytes, 0, d_b, r_b, b.nbytes)
    np.testing.assert_array_equal(r_a.download(a.shape, np.uint16), a)
    np.testing.assert_array_equal(r_b.download(b.shape, np.float32), b)
    comm.halo_exchange(-1, None, None, 0, -1, None, None, 0)            # no neighbours: nothing to do
    assert comm.max(3.5) == 3.5
    with pytest.raises(ValueError):
        comm.halo_exchange(2, d_a, r_a, 8, -1, None, None, 0)          # peer outside the communicator
    for buf in (d_a, d_b, r_a, r_b):
        buf.free()


def test_native_slab_and_chunk_drivers_with_one_rank(ctx, comm):
    """world = 1: no neighbour, the drivers reduce to the one-call pipelines -- the same uint16 volumes."""
    from aind_exaspim_image_compression.bm4d import denoise_chunked, denoise_volume
    vol, _ = synth_volume((48, 40, 44), seed=43, as_u16=True)
    plan = D.plan_slabs(48, 1, 0)
    d_raw = ctx.to_device(vol)
    out = D.denoise_slab_u16_native(ctx, comm, d_raw, plan, vol.shape, SIGMA, OFFSET)
    np.testing.assert_array_equal(out.download(vol.shape, np.uint16), denoise_volume(vol, SIGMA, OFFSET))
    out.free()
    cplan = D.plan_chunk_slabs(48, 1, 0, chunk=16, halo=8)
    out = D.denoise_chunked_slab_native(ctx, comm, d_raw, cplan, vol.shape, SIGMA, OFFSET, chunk=16, halo=8)
    np.testing.assert_array_equal(out.download(vol.shape, np.uint16), denoise_chunked(vol, SIGMA, OFFSET, chunk=16, halo=8))
    out.free()
    d_raw.free()
    c2 = D.rendezvous_comm(ctx, 0, 1, tag="test-one-rank")             # the file rendezvous, trivially
    assert c2.max(1.0) == 1.0
    c2.close()


@pytest.mark.parametrize("mode", ["slabs", "chunks"])
def test_bench_native_comm_mode_runs_without_torch(mode):
    """`bench.py --mode slabs|chunks --comm native` (one rank): the line, and torch never imported."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", mode, "--comm", "native", "--size", "64",
                          "--chunk", "32", "--steps", "1", "--warmup", "1", "--bm4dnet", "0", "--cpu-sample", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["config"]["comm"] == "native" and d["value"] > 0
    assert 15.0 < d["residual_std"] < 30.0


@pytest.mark.skipif(_native.device_count_no_init() < 2, reason="two ranks of RCCL need two devices")
def test_two_ranks_native_halo_exchange(tmp_path):
    """Two devices: `bench.py --gpus 2 --mode slabs --comm native` through its own launcher."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "slabs", "--comm", "native",
                          "--size", "64", "--steps", "1", "--warmup", "1", "--bm4dnet", "0", "--cpu-sample", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 2 and d["config"]["comm"] == "native"
