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


def plan_slabs(nz, world, rank, halo=EXACT_HALO, align=4):
    """Split ``nz`` planes into ``world`` contiguous slabs whose boundaries are multiples of
    ``align``; every slab gets at least ``halo`` planes so that a halo never spans two ranks."""
    if halo % align:
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


def exchange_basic_halo(basic, plan, dist=None, group=None):
    """Fill the halo planes of ``basic`` (a torch tensor [p1-p0, ny, nx] on this rank's device)
    with the neighbours' owned planes, in place.  Blocking; all ranks must call it."""
    if plan.world == 1:
        return basic
    import torch
    if dist is None:
        import torch.distributed as dist
    ops, keep = [], []
    core = plan.core
    n_own = plan.z1 - plan.z0
    if plan.rank > 0:
        send = basic[core.start:core.start + min(plan.halo, n_own)].contiguous()
        recv = torch.empty_like(basic[:plan.lo])
        ops += [dist.P2POp(dist.isend, send, plan.rank - 1, group),
                dist.P2POp(dist.irecv, recv, plan.rank - 1, group)]
        keep.append((slice(0, plan.lo), recv))
    if plan.rank < plan.world - 1:
        send = basic[core.stop - min(plan.halo, n_own):core.stop].contiguous()
        recv = torch.empty_like(basic[core.stop:])
        ops += [dist.P2POp(dist.isend, send, plan.rank + 1, group),
                dist.P2POp(dist.irecv, recv, plan.rank + 1, group)]
        keep.append((slice(core.stop, basic.shape[0]), recv))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for sl, buf in keep:
        basic[sl].copy_(buf)
    return basic


def denoise_slab(noisy, plan, sigma, stage1, stage2, dist=None, group=None):
    """Two-stage BM4D of this rank's padded slab ``noisy`` ([p1-p0, ny, nx] fp32 tensor).

    ``stage1(noisy) -> basic`` and ``stage2(noisy, basic) -> estimate`` operate on whole padded
    slabs (on the GPU they are ``SlabDenoiser.stage1/stage2``).  Returns the estimate of the
    owned planes only."""
    basic = stage1(noisy)
    exchange_basic_halo(basic, plan, dist=dist, group=group)
    out = stage2(noisy, basic)
    return out[plan.core]


class SlabDenoiser:
    """The two stage callables of ``denoise_slab`` on one MI355X, through the staged C-ABI entry
    points (exabm4d_blockmatch_dev / exabm4d_stage_dev / exabm4d_normalize_dev), with all
    scratch held as torch tensors on the rank's device."""

    def __init__(self, shape, sigma, device, params=None):
        import torch
        from aind_exaspim_image_compression import _native
        self.torch = torch
        self.shape = tuple(int(s) for s in shape)
        self.sigma = float(sigma)
        self.params = params or _native.default_params()
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
            self.num.zero_()
            self.den.zero_()
            ctx.blockmatch(match_on, self.shape, self.sigma, c_match, self.keys, self.params)
            ctx.stage(noisy, basic, self.keys, self.shape, self.sigma, self.num, self.den,
                      self.params)
            out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
            ctx.normalize(self.num, self.den, out, n)
            stream.synchronize()
            ctx.reset_stream()
        return out

    def stage1(self, noisy):
        return self._run(noisy, self.params.c_match_ht, noisy, None)

    def stage2(self, noisy, basic):
        return self._run(basic, self.params.c_match_wie, noisy, basic)
