"""Multi-GPU BM4D: one process per GPU, z-slabs, halo exchange of the basic estimate.

The reference has no multi-device code at all; its precedent is that
patches / chunks are independent (scripts/precompute.py:215-228).  For one large volume the only
coupling between z-slabs is the stage-1 *basic estimate* that stage 2 reads in the halo of a
slab: each rank sends the outer ``halo`` z-planes of the basic estimate it owns to its two slab
neighbours (point-to-point ``isend``/``irecv`` -- RCCL over xGMI with the ``nccl`` backend, gloo in
the CPU tests).  There is no all-reduce and no gather: every rank keeps its own output slab.

Semantics.  Rank r owns planes [z0, z1) (multiples of 4, so the reference-block grid of the slab
coincides with the grid of the whole volume).  It reads the noisy input on [z0-halo, z1+halo)
(clamped to the volume), runs stage 1 there, keeps the basic estimate of its own planes, receives
its neighbours' basic estimate for the halo planes, runs stage 2 and writes [z0, z1).  A voxel's
stage output depends on input within 24 voxels (blocks of groups whose reference lies within 12,
whose candidates lie within another 12), so with ``halo = 24`` the sharded result equals the
whole-volume result up to the fp32 summation order; ``halo = 8`` (BASELINE.json config 4) is the
cheaper chunk-local approximation.
"""
from dataclasses import dataclass

EXACT_HALO = 24


def offset_exact_in_fp32(offset):
    """(float)v - offset is exact for every uint16 v iff the offset has at most 7 fractional bits;
    only then does matching on the uint16 planes give the tables of matching on the fp32 counts
    (csrc/exabm4d_api.hip: offset_exact_in_fp32)."""
    import numpy as np
    off = np.float32(offset)
    s = off * np.float32(128.0)
    return bool(abs(off) <= 65536.0 and s == np.rint(s))


@dataclass(frozen=True)
class SlabPlan:
    rank: int
    world: int
    nz: int        # planes of the whole volume
    z0: int        # owned planes [z0, z1)
    z1: int
    p0: int        # padded planes [p0, p1) actually held by this rank
    p1: int
    halo: int

    @property
    def lo(self):
        """planes of halo below the owned range that exist (0 for the first slab)"""
        return self.z0 - self.p0

    @property
    def hi(self):
        return self.p1 - self.z1

    @property
    def core(self):
        """slice of the owned planes inside the padded slab"""
        return slice(self.lo, self.lo + (self.z1 - self.z0))


def plan_slabs(nz, world, rank, halo=EXACT_HALO, align=4, halo_step=4):
    """Split ``nz`` planes into ``world`` contiguous slabs whose boundaries are multiples of
    ``align``; every slab gets at least ``halo`` planes so that a halo never spans two ranks.
    (Chunk-local mode: ``align`` = the chunk edge, so that no chunk straddles two ranks.)"""
    if halo % halo_step:
        raise ValueError("halo must be a multiple of the grid step")
    units = nz // align
    if units < world:
        raise ValueError(f"volume too thin: {nz} planes for {world} ranks")
    bounds = [align * ((units * r) // world) for r in range(world)] + [nz]
    z0, z1 = bounds[rank], bounds[rank + 1]
    if world > 1 and min(bounds[r + 1] - bounds[r] for r in range(world)) < halo:
        raise ValueError(f"slabs thinner than the halo ({halo}); use fewer ranks")
    return SlabPlan(rank=rank, world=world, nz=nz, z0=z0, z1=z1, p0=max(0, z0 - halo),
                    p1=min(nz, z1 + halo), halo=halo)


class HaloExchange:
    """An in-flight exchange of halo planes with the two slab neighbours: batched point-to-point
    ``isend``/``irecv`` (RCCL over xGMI under the ``nccl`` backend).  ``wait()`` completes it and
    copies the received planes into the halo of ``slab``.  Under the ``gloo`` backend (CPU tests,
    single-GPU rehearsals) device tensors are staged through host memory."""

    def __init__(self, slab, plan, dist=None, group=None):
        import torch
        self.slab, self.reqs, self.keep, self.ops = slab, [], [], []
        if plan.world == 1:
            return
        if dist is None:
            import torch.distributed as dist
        stage = slab.is_cuda and dist.get_backend(group) == "gloo"

        # 16-bit integer planes (raw counts) travel as bytes: RCCL has no int16 element type
        as_bytes = slab.dtype not in (torch.float32, torch.float64, torch.float16, torch.bfloat16,
                                      torch.int32, torch.int64, torch.uint8, torch.int8)

        def out(t):
            t = t.contiguous()
            if as_bytes:
                t = t.view(torch.uint8)
            return t.cpu() if stage else t

        def inbox(like):
            shape = like.shape[:-1] + (like.shape[-1] * like.element_size(),) if as_bytes else like.shape
            return torch.empty(shape, dtype=torch.uint8 if as_bytes else like.dtype,
                               device="cpu" if stage else like.device)

        ops = []
        core = plan.core
        n_own = plan.z1 - plan.z0
        if plan.rank > 0:
            recv = inbox(slab[:plan.lo])
            ops += [dist.P2POp(dist.isend, out(slab[core.start:core.start + min(plan.halo, n_own)]),
                               plan.rank - 1, group),
                    dist.P2POp(dist.irecv, recv, plan.rank - 1, group)]
            self.keep.append((slice(0, plan.lo), recv))
        if plan.rank < plan.world - 1:
            recv = inbox(slab[core.stop:])
            ops += [dist.P2POp(dist.isend, out(slab[core.stop - min(plan.halo, n_own):core.stop]),
                               plan.rank + 1, group),
                    dist.P2POp(dist.irecv, recv, plan.rank + 1, group)]
            self.keep.append((slice(core.stop, slab.shape[0]), recv))
        self.ops = ops                                  # keeps the send buffers alive
        self.reqs = dist.batch_isend_irecv(ops)

    def wait(self):
        for req in self.reqs:
            req.wait()
        for sl, buf in self.keep:
            if buf.dtype != self.slab.dtype:
                buf = buf.view(self.slab.dtype)
            self.slab[sl].copy_(buf)
        self.reqs, self.keep, self.ops = [], [], []
        return self.slab


def exchange_basic_halo(basic, plan, dist=None, group=None):
    """Fill the halo planes of ``basic`` (a torch tensor [p1-p0, ny, nx] on this rank's device)
    with the neighbours' owned planes, in place.  Blocking; all ranks must call it."""
    return HaloExchange(basic, plan, dist=dist, group=group).wait()


def denoise_slab(noisy, plan, sigma, stage1, stage2, dist=None, group=None):
    """Two-stage BM4D of this rank's padded slab ``noisy`` ([p1-p0, ny, nx] fp32 tensor).

    ``stage1(noisy) -> basic`` and ``stage2(noisy, basic) -> estimate`` operate on whole padded
    slabs (on the GPU they are ``SlabDenoiser.stage1/stage2``).  Returns the estimate of the
    owned planes only."""
    basic = stage1(noisy)
    exchange_basic_halo(basic, plan, dist=dist, group=group)
    out = stage2(noisy, basic)
    return out[plan.core]


def denoise_slab_u16(raw, plan, offset, denoiser, dist=None, group=None):
    """The uint16 pipeline of this rank's padded slab ``raw`` ([p1-p0, ny, nx] int16 view of the
    counts) with the fused ends of ``exabm4d_denoise_u16_dev``: ``denoiser`` is a ``SlabDenoiser``.
    Returns the owned planes (int16 view of uint16)."""
    noisy, basic = denoiser.stage1_u16(raw, offset)
    exchange_basic_halo(basic, plan, dist=dist, group=group)
    return denoiser.stage2_u16(noisy, basic, offset)[plan.core]


def global_data_exp(noisy, dist=None, group=None):
    """E with max |v| < 2^E over the WHOLE sharded fp32 volume: the exponent of this
    rank's largest |v| bit pattern, MAX-reduced over the ranks (one int; the only collective the exact
    slab mode adds to its halo exchange).  ``noisy``: this rank's planes as a torch tensor."""
    import torch
    bits = int(noisy.detach().abs().max().view(torch.int32).item()) if noisy.numel() else 0
    e = torch.tensor([(bits >> 23) - 126], dtype=torch.int32)
    if dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) != "gloo":          # RCCL reduces device tensors, gloo host tensors
            e = e.to(noisy.device)
        dist.all_reduce(e, op=dist.ReduceOp.MAX, group=group)
    return int(e.item())


class SlabDenoiser:
    """The two stage callables of ``denoise_slab`` on one MI355X, through the staged C-ABI entry
    points (exabm4d_blockmatch_dev / exabm4d_stage_dev / exabm4d_normalize_dev), with all
    scratch held as torch tensors on the rank's device."""

    def __init__(self, shape, sigma, device, params=None, data_exp=None):
        """``data_exp``: E of the numerator's fixed-point unit for the fp32 callables
        ``stage1`` / ``stage2``.  The single-GPU fp32 pipeline reads E off the whole volume; a sharded run
        that wants the same bits passes the same E to every rank (``global_data_exp``).  None: from this
        rank's slab.  The uint16 callables always use the uint16 pipelines' E = 17."""
        import torch
        from aind_exaspim_image_compression import _native
        self.torch = torch
        self.shape = tuple(int(s) for s in shape)
        self.sigma = float(sigma)
        self.params = params or _native.default_params()
        self.data_exp = data_exp
        self.u16_exp = _native.DATA_EXP_U16
        self.device = torch.device(device)
        self.ctx = _native.context(self.device.index or 0)
        g = [len(_native.grid_positions(n)) for n in self.shape]
        self.keys = torch.empty((g[0], g[1], g[2], 16), dtype=torch.int32, device=self.device)
        self.num = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        self.den = torch.empty(self.shape, dtype=torch.float32, device=self.device)

    def _run(self, match_on, c_match, noisy, basic):
        torch, ctx = self.torch, self.ctx
        n = noisy.numel()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            ctx.set_stream(stream.cuda_stream)
            ctx.blockmatch(match_on, self.shape, self.sigma, c_match, self.keys, self.params)
            ctx.stage(noisy, basic, self.keys, self.shape, self.sigma, self.num, self.den,
                      self.params, data_exp=self.data_exp)
            out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
            ctx.normalize(self.num, self.den, out, n)
            stream.synchronize()
            ctx.reset_stream()
        return out

    def stage1(self, noisy):
        return self._run(noisy, self.params.c_match_ht, noisy, None)

    def stage2(self, noisy, basic):
        return self._run(basic, self.params.c_match_wie, noisy, basic)

    # ---- the uint16 pipeline of exabm4d_denoise_u16_dev, split at the halo exchange ---------------
    def stage1_u16(self, raw, offset):
        """``raw``: the padded slab's counts (an int16 view of the uint16 planes).  Returns
        (noisy, basic): fp32 counts minus ``offset`` and the stage-1 estimate, both whole padded
        slabs.  Matching runs on the uint16 planes themselves (integer kernel where it applies)."""
        torch, ctx = self.torch, self.ctx
        n = raw.numel()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            ctx.set_stream(stream.cuda_stream)
            noisy = torch.empty(self.shape, dtype=torch.float32, device=self.device)
            basic = torch.empty(self.shape, dtype=torch.float32, device=self.device)
            ctx.counts_from_u16(raw, noisy, n, float(offset))
            if offset_exact_in_fp32(offset):
                ctx.blockmatch_u16(raw, self.shape, self.sigma, self.params.c_match_ht, self.keys,
                                   self.params)
            else:                       # (float)v - offset is rounded: match on what stage 1 filters
                ctx.blockmatch(noisy, self.shape, self.sigma, self.params.c_match_ht, self.keys,
                               self.params)
            ctx.stage(noisy, None, self.keys, self.shape, self.sigma, self.num, self.den, self.params,
                      data_exp=self.u16_exp)
            ctx.normalize(self.num, self.den, basic, n)
            stream.synchronize()
            ctx.reset_stream()
        return noisy, basic

    def stage2_u16(self, noisy, basic, offset):
        """Stage 2 on the padded slab and the uint16 cast (+ offset, clamp, rint) of every plane;
        returns an int16 view tensor like ``raw``."""
        torch, ctx = self.torch, self.ctx
        n = noisy.numel()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            ctx.set_stream(stream.cuda_stream)
            # the uint16 pipelines match stage 2 on the basic estimate rounded to counts;
            # self.den is free until the stage call writes it
            ctx.round_counts(basic, self.den, n, float(offset))
            ctx.blockmatch(self.den, self.shape, self.sigma, self.params.c_match_wie, self.keys,
                           self.params)
            ctx.stage(noisy, basic, self.keys, self.shape, self.sigma, self.num, self.den, self.params,
                      data_exp=self.u16_exp)
            out = torch.empty(self.shape, dtype=torch.int16, device=self.device)
            ctx.normalize_u16(self.num, self.den, out, n, float(offset))
            stream.synchronize()
            ctx.reset_stream()
        return out


# ---- chunk-local mode across ranks (BASELINE.json config 4) -------------------------------------
def plan_chunk_slabs(nz, world, rank, chunk=256, halo=8):
    """z-slabs whose boundaries are multiples of ``chunk``: every rank owns whole layers of chunks
    and needs only ``halo`` raw input planes from each neighbour."""
    return plan_slabs(nz, world, rank, halo=halo, align=chunk, halo_step=1)


def denoise_chunked_slab(raw, plan, run_chunks, chunk=256, dist=None, group=None):
    """Chunk-local BM4D of this rank's slab.  ``raw``: [p1-p0, ny, nx] tensor of the raw counts (an
    int16 view of the uint16 planes) whose owned planes are filled in; the neighbours' ``halo``
    input planes are exchanged here, *while the chunk layers that do not touch them are already
    being denoised*.  ``run_chunks(raw, (zc0, zc1)) -> core planes`` processes the chunk layers
    whose cores are the local planes [zc0, zc1) (on the GPU: ``ChunkedSlabDenoiser.run``).  Returns
    the owned planes."""
    import torch
    ex = HaloExchange(raw, plan, dist=dist, group=group)           # in flight from here on
    core = plan.core
    n_own = core.stop - core.start
    lo_edge = core.start + (chunk if plan.lo else 0)               # first layer needs the lower halo
    hi_edge = core.stop - (chunk if plan.hi else 0)
    if lo_edge >= hi_edge:                                          # one or two layers: nothing to overlap
        ex.wait()
        return run_chunks(raw, (core.start, core.stop))
    parts = {}
    parts[lo_edge] = run_chunks(raw, (lo_edge, hi_edge))           # interior layers: own planes only
    ex.wait()
    if plan.lo:
        parts[core.start] = run_chunks(raw, (core.start, lo_edge))
    if plan.hi:
        parts[hi_edge] = run_chunks(raw, (hi_edge, core.stop))
    out = torch.cat([parts[k] for k in sorted(parts)], dim=0)
    assert out.shape[0] == n_own
    return out


class ChunkedSlabDenoiser:
    """``run_chunks`` of ``denoise_chunked_slab`` on one MI355X: one batched
    ``exabm4d_denoise_chunked_u16_dev`` call per core range, torch tensors in and out."""

    def __init__(self, sigma, offset, device, chunk=256, halo=8, params=None):
        import torch
        from aind_exaspim_image_compression import _native
        self.torch = torch
        self.sigma, self.offset, self.chunk, self.halo = float(sigma), float(offset), int(chunk), int(halo)
        self.params = params or _native.default_params()
        self.device = torch.device(device)
        self.ctx = _native.context(self.device.index or 0)

    def run(self, raw, core):
        torch, ctx = self.torch, self.ctx
        shape = tuple(int(s) for s in raw.shape)
        out = torch.empty((core[1] - core[0],) + shape[1:], dtype=torch.int16, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            ctx.set_stream(stream.cuda_stream)
            ctx.denoise_chunked_u16(raw, out, shape, self.sigma, self.offset, chunk=self.chunk,
                                    halo=self.halo, core=core, params=self.params)
            stream.synchronize()
            ctx.reset_stream()
        return out


# ---- the same two modes without torch: RCCL behind the C-ABI (round 4) ---------------------------------------
# north_star: "PyTorch-ROCm used only for the bm4dnet stage ... RCCL over xGMI only for halo exchange".  The
# classes above keep their slabs in torch tensors and exchange through torch.distributed (what the gloo tests
# and the single-GPU rehearsals run); the functions below need neither: DeviceBuffers, the context's stream,
# and exabm4d_halo_exchange_dev (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, csrc/comm_rccl.hip).
def halo_messages(plan, plane_bytes, what="basic"):
    """Byte offsets inside a padded slab buffer [p1 - p0 planes] of the four messages of one exchange:
    -> ((lo_peer, send_lo_off, recv_lo_off, bytes_lo), (hi_peer, send_hi_off, recv_hi_off, bytes_hi)); a peer of
    -1 = no neighbour on that side.  A rank sends the outer ``halo`` planes it OWNS and receives its
    neighbour's into its halo region; the planes of a slab are contiguous, so these are plain ranges."""
    core = plan.core
    n_own = plan.z1 - plan.z0
    h = min(plan.halo, n_own)
    lo = (-1, 0, 0, 0)
    hi = (-1, 0, 0, 0)
    if plan.rank > 0:
        lo = (plan.rank - 1, core.start * plane_bytes, 0, plan.lo * plane_bytes)
        if plan.lo != h:
            raise ValueError("slab thinner than its neighbour's halo")
    if plan.rank < plan.world - 1:
        hi = (plan.rank + 1, (core.stop - h) * plane_bytes, core.stop * plane_bytes, plan.hi * plane_bytes)
        if plan.hi != h:
            raise ValueError("slab thinner than its neighbour's halo")
    return lo, hi


def exchange_halo_native(comm, buf, plan, plane_bytes):
    """One halo exchange of the padded slab ``buf`` (DeviceBuffer / device pointer) on the context's stream."""
    if plan.world == 1:
        return
    from aind_exaspim_image_compression import _native
    base = _native._ptr(buf)
    (lp, ls, lr, lb), (hp, hs, hr, hb) = halo_messages(plan, plane_bytes)
    comm.halo_exchange(lp, base + ls, base + lr, lb, hp, base + hs, base + hr, hb)


def rendezvous_comm(ctx, rank, world, tag=None, timeout=180.0):
    """A ``_native.Comm`` over all ranks of a single-node job, without torch: rank 0 draws the RCCL unique id
    and publishes it in a file the other ranks poll for (same node, so the temp directory is shared; the name
    carries the launcher's pid and MASTER_PORT / TORCHELASTIC_RUN_ID, so that two jobs do not meet)."""
    import os
    import tempfile
    import time
    from aind_exaspim_image_compression import _native
    if tag is None:
        tag = "%s-%s-%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "run"), os.environ.get("MASTER_PORT", "0"),
                            os.getppid())
    path = os.path.join(tempfile.gettempdir(), f"exabm4d-comm-{os.getuid()}-{tag}.id")
    if rank == 0:
        uid = _native.comm_unique_id()
        fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        uid = b""
        while len(uid) != _native.COMM_ID_BYTES:
            try:
                with open(path, "rb") as f:
                    uid = f.read()
            except FileNotFoundError:
                pass
            if len(uid) != _native.COMM_ID_BYTES:
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"rank {rank}: no RCCL id from rank 0 in {path}")
                time.sleep(0.02)
    comm = _native.Comm(ctx, world, rank, uid)          # collective: returns once every rank has the id
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
    return comm


def denoise_slab_u16_native(ctx, comm, d_raw, plan, shape, sigma, offset, params=None):
    """The exact slab mode of ``denoise_slab_u16`` on DeviceBuffers: ``d_raw`` holds this rank's padded slab of
    uint16 counts ([p1 - p0, ny, nx]); returns a DeviceBuffer with the padded slab's uint16 result (the owned
    planes are ``plan.core``).  Stage 1 on the padded slab, the basic estimate's halo from the neighbours
    through ``comm`` (fp32 planes as bytes), stage 2, the uint16 cast -- everything on the context's stream,
    no host synchronisation inside."""
    from aind_exaspim_image_compression import _native
    p = params or _native.default_params()
    shape = tuple(int(s) for s in shape)
    n = int(shape[0]) * shape[1] * shape[2]
    g = [len(_native.grid_positions(m)) for m in shape]
    d_noisy, d_basic = ctx.alloc(4 * n), ctx.alloc(4 * n)
    d_num, d_den = ctx.alloc(4 * n), ctx.alloc(4 * n)
    d_keys = ctx.alloc(g[0] * g[1] * g[2] * 64)
    d_out = ctx.alloc(2 * n)
    try:
        ctx.counts_from_u16(d_raw, d_noisy, n, float(offset))
        if offset_exact_in_fp32(offset):
            ctx.blockmatch_u16(d_raw, shape, sigma, p.c_match_ht, d_keys, p)
        else:
            ctx.blockmatch(d_noisy, shape, sigma, p.c_match_ht, d_keys, p)
        ctx.stage(d_noisy, None, d_keys, shape, sigma, d_num, d_den, p, data_exp=_native.DATA_EXP_U16)
        ctx.normalize(d_num, d_den, d_basic, n)
        exchange_halo_native(comm, d_basic, plan, 4 * shape[1] * shape[2])
        ctx.round_counts(d_basic, d_den, n, float(offset))          # stage 2 matches on counts
        ctx.blockmatch(d_den, shape, sigma, p.c_match_wie, d_keys, p)
        ctx.stage(d_noisy, d_basic, d_keys, shape, sigma, d_num, d_den, p, data_exp=_native.DATA_EXP_U16)
        ctx.normalize_u16(d_num, d_den, d_out, n, float(offset))
        return d_out
    finally:
        ctx.sync()
        for b in (d_noisy, d_basic, d_num, d_den, d_keys):
            b.free()


def denoise_chunked_slab_native(ctx, comm, d_raw, plan, shape, sigma, offset, chunk=256, halo=8, params=None):
    """Chunk-local mode (BASELINE config 4) of this rank's slab on DeviceBuffers: the neighbours' ``halo`` raw
    planes arrive through ``comm`` (uint16 planes as bytes), then one batched
    ``exabm4d_denoise_chunked_u16_dev`` call over the owned chunk layers.  Returns a DeviceBuffer with the
    owned planes.  (The torch path overlaps the exchange with the interior layers; here it is simply ordered
    on the one stream in front of the kernels: 8 planes are 0.5 % of the data a rank reads.)"""
    shape = tuple(int(s) for s in shape)
    exchange_halo_native(comm, d_raw, plan, 2 * shape[1] * shape[2])
    core = plan.core
    d_out = ctx.alloc(2 * (core.stop - core.start) * shape[1] * shape[2])
    ctx.denoise_chunked_u16(d_raw, d_out, shape, float(sigma), float(offset), chunk=int(chunk), halo=int(halo),
                            core=(core.start, core.stop), params=params)
    return d_out
