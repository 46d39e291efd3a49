"""N > 1 path with the HIP kernels as the compute: two ranks share cuda:0 (rehearsal-style: gloo
rendezvous, device tensors staged through the host in the halo exchange) and run
(a) the exact slab mode -- SlabDenoiser stage callables, basic-estimate halo exchange -- and
(b) the chunk-local mode of BASELINE.json config 4 -- raw input halo exchange overlapped with the
interior chunk layers, ChunkedSlabDenoiser -- and the stitched results must equal the
single-process device results; (c) the uint16 form of the slab mode (fused conversion ends,
integer matching on the uint16 planes) against exabm4d_denoise_u16_dev of the whole volume."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import synth_volume

pytestmark = pytest.mark.gpu
SIGMA, OFFSET = 24.0, 37.0
SLAB_SHAPE = (96, 40, 44)
CHUNK_SHAPE = (96, 40, 48)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmp):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "aind-exaspim-image-compression_amd"),
              os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from aind_exaspim_image_compression.distributed import (ChunkedSlabDenoiser, SlabDenoiser,
                                                            denoise_chunked_slab, denoise_slab,
                                                            denoise_slab_u16, global_data_exp,
                                                            plan_chunk_slabs, plan_slabs)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    dev = torch.device("cuda", 0)
    # (a) exact slabs
    vol, _ = synth_volume(SLAB_SHAPE, seed=41)
    plan = plan_slabs(SLAB_SHAPE[0], world, rank)
    noisy = torch.from_numpy(np.ascontiguousarray(vol[plan.p0:plan.p1])).to(dev)
    # the numerator's unit of the fp32 pipeline is read off the WHOLE volume: one MAX-reduced int
    den = SlabDenoiser(tuple(noisy.shape), SIGMA, dev, data_exp=global_data_exp(noisy[plan.core], dist=dist))
    out = denoise_slab(noisy, plan, SIGMA, den.stage1, den.stage2)
    np.save(os.path.join(tmp, f"slab{rank}.npy"), out.cpu().numpy())
    # (b) chunk-local slabs: only the owned raw planes are filled in before the exchange
    raw_np, _ = synth_volume(CHUNK_SHAPE, seed=42, as_u16=True)
    cplan = plan_chunk_slabs(CHUNK_SHAPE[0], world, rank, chunk=16, halo=8)
    raw = torch.zeros((cplan.p1 - cplan.p0,) + CHUNK_SHAPE[1:], dtype=torch.int16, device=dev)
    raw[cplan.core] = torch.from_numpy(raw_np[cplan.z0:cplan.z1].view(np.int16)).to(dev)
    cden = ChunkedSlabDenoiser(SIGMA, OFFSET, dev, chunk=16, halo=8)
    cout = denoise_chunked_slab(raw, cplan, cden.run, chunk=16)
    np.save(os.path.join(tmp, f"cslab{rank}.npy"), cout.cpu().numpy().view(np.uint16))
    # (c) exact slabs, uint16 pipeline
    u_np, _ = synth_volume(SLAB_SHAPE, seed=43, as_u16=True)
    uraw = torch.from_numpy(np.ascontiguousarray(u_np[plan.p0:plan.p1]).view(np.int16)).to(dev)
    uout = denoise_slab_u16(uraw, plan, OFFSET, den)
    np.save(os.path.join(tmp, f"uslab{rank}.npy"), uout.cpu().numpy().view(np.uint16))
    np.save(os.path.join(tmp, f"plans{rank}.npy"), np.array([plan.z0, plan.z1, cplan.z0, cplan.z1]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_hip_compute(ctx, tmp_path):
    from aind_exaspim_image_compression.bm4d import denoise_chunked
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    vol, _ = synth_volume(SLAB_SHAPE, seed=41)
    whole = ctx.denoise_f32_host(vol, SIGMA, stages=2)
    raw_np, _ = synth_volume(CHUNK_SHAPE, seed=42, as_u16=True)
    cwhole = denoise_chunked(raw_np, SIGMA, OFFSET, chunk=16, halo=8)
    u_np, _ = synth_volume(SLAB_SHAPE, seed=43, as_u16=True)
    d_in, d_out = ctx.to_device(u_np), ctx.alloc(u_np.nbytes)
    ctx.denoise_u16(d_in, d_out, SLAB_SHAPE, SIGMA, OFFSET)
    uwhole = d_out.download(SLAB_SHAPE, np.uint16)
    got, cgot, ugot = np.empty_like(whole), np.empty_like(cwhole), np.empty_like(uwhole)
    for r in range(world):
        z0, z1, c0, c1 = np.load(tmp_path / f"plans{r}.npy")
        got[z0:z1] = np.load(tmp_path / f"slab{r}.npy")
        cgot[c0:c1] = np.load(tmp_path / f"cslab{r}.npy")
        ugot[z0:z1] = np.load(tmp_path / f"uslab{r}.npy")
    # 24-plane halo, integer aggregation sums: the sharded results ARE the single-process results
    np.testing.assert_array_equal(got, whole)
    np.testing.assert_array_equal(cgot, cwhole)
    np.testing.assert_array_equal(ugot, uwhole)
