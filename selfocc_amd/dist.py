"""Multi-GPU layer of the hot path: one process per GPU, ``torch.distributed`` with backend
"nccl" (= RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).

Rays are independent units and the (small) field volume is replicated, so the path shards
with NO data-path collective:
  * ``shard_rays``     contiguous row blocks of every camera's ray lattice per rank (keeps the
                       image locality the 8x8 wavefront tiles and the SSIM windows rely on);
  * eval:              every rank scores its shard, the integer / float metric sums are
                       all-reduced once per epoch (MeanIoU._after_epoch), or the rendered maps
                       are ``gather_rays``-ed;
  * train:             ``replicate_grad_sum`` marks the replicated field volume: identity in
                       forward, ONE all-reduce(sum) of dL/d(volume) in backward (the exchange step
                       of SURVEY §8e cfg3); every rank renders its row block and
                       ``gather_rays_autograd`` rebuilds the full-frame PER-RAY tensors the
                       (lattice-shaped) losses consume — its backward hands each rank the slice of the
                       gradient that belongs to its rays.  The PER-SAMPLE tensors (weights / ts / deltas /
                       sdf / eik_grad: 206 MB at the shipped training size) stay local: the head hands
                       them over as ``LocalRows`` / tagged tensors and the losses that consume them
                       reduce them to per-ray terms on the local rows first (``loss/reproj.py``,
                       ``EikonalLoss``), gathering 5 floats per ray instead.  ``all_reduce_mean`` reports
                       the global rendered-depth loss.
``NeuSHead(ray_shard=True)`` (or SELFOCC_RAY_SHARD=1) switches the head to this mode; bench.py
``--shard rays`` times it.  The reference itself only has frame-per-GPU DDP (train.py:86-91).
"""
import torch
import torch.distributed as dist

from .render import RaySet


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def row_block(n_rows, rank, world_size):
    """[r0, r1) of ``rank``: sizes differ by at most one row, earlier ranks get the extras."""
    base, extra = divmod(n_rows, world_size)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def shard_rays(rays: RaySet, rank=None, world_size=None) -> RaySet:
    """The rays of ``rank``.  Pixel-lattice rays: rows [r0, r1) of every camera (the shard is
    again a lattice, so in-kernel ray generation and tiling still apply).  Explicit rays:
    a contiguous slice."""
    if rank is None:
        rank, world_size = world()
    if world_size == 1:
        return rays
    if rays.pixel_grid:
        r0, r1 = row_block(rays.ny, rank, world_size)
        return RaySet(img2lidar=rays.img2lidar, nx=rays.nx, ny=r1 - r0, sx=rays.sx, sy=rays.sy, ox=rays.ox,
                      oy=rays.oy + r0 * rays.sy)
    s0, s1 = row_block(rays.n_rays, rank, world_size)
    return RaySet(origins=rays.origins[s0:s1].contiguous(), dirs=rays.dirs[s0:s1].contiguous(),
                  dir_norm=None if rays.dir_norm is None else rays.dir_norm[s0:s1].contiguous())


def _all_gather(bufs, t):
    """dist.all_gather; gloo has no CUDA all_gather, so CUDA tensors are staged through the host there (only the
    single-GPU test rigs run gloo on CUDA tensors — production is RCCL)."""
    if t.is_cuda and dist.get_backend() == 'gloo':
        host = [torch.empty(b.shape, dtype=b.dtype) for b in bufs]
        dist.all_gather(host, t.cpu())
        for b, h in zip(bufs, host):
            b.copy_(h)
        return
    dist.all_gather(bufs, t)


def gather_rays(t, rays: RaySet, dim_per_ray=()):
    """All-gather a per-ray tensor of the local shard back into full-frame order
    (n_cams, ny, nx, ...) for lattices / (n_rays, ...) for explicit rays.  ``rays`` is the
    UNSHARDED RaySet."""
    rank, ws = world()
    if ws == 1:
        return t
    if rays.pixel_grid:
        n_cams = rays.img2lidar.shape[0]
        rows = [row_block(rays.ny, r, ws) for r in range(ws)]
        max_rows = max(b - a for a, b in rows)
        loc = t.reshape(n_cams, rows[rank][1] - rows[rank][0], rays.nx, *t.shape[1:])
        pad = loc.new_zeros(n_cams, max_rows, rays.nx, *t.shape[1:])
        pad[:, :loc.shape[1]] = loc
        bufs = [torch.empty_like(pad) for _ in range(ws)]
        _all_gather(bufs, pad.contiguous())
        return torch.cat([b[:, :(r1 - r0)] for b, (r0, r1) in zip(bufs, rows)], dim=1)
    sizes = [row_block(rays.n_rays, r, ws) for r in range(ws)]
    mx = max(b - a for a, b in sizes)
    pad = t.new_zeros(mx, *t.shape[1:])
    pad[:t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    _all_gather(bufs, pad.contiguous())
    return torch.cat([b[:(s1 - s0)] for b, (s0, s1) in zip(bufs, sizes)], dim=0)


def all_reduce_mean(value: torch.Tensor, count):
    """Global mean of a per-rank mean over ``count`` local units (the rendered-depth loss of
    north_star): sum(value * count) / sum(count) with ONE all-reduce of a 2-vector."""
    rank, ws = world()
    if ws == 1:
        return value
    buf = torch.stack([value.detach().reshape(()) * count, value.new_tensor(float(count))])
    dist.all_reduce(buf)
    return buf[0] / buf[1]


def local_slice(full, rays: RaySet, rank=None, world_size=None):
    """Inverse of ``gather_rays`` for one rank: the rows of a full-frame per-ray tensor ((n_cams * ny * nx, ...)
    or (n_rays, ...)) that belong to ``rank``'s shard, flattened back to (n_local, ...)."""
    if rank is None:
        rank, world_size = world()
    if world_size == 1:
        return full
    if rays.pixel_grid:
        n_cams = rays.img2lidar.shape[0]
        r0, r1 = row_block(rays.ny, rank, world_size)
        f = full.reshape(n_cams, rays.ny, rays.nx, *full.shape[1:])
        return f[:, r0:r1].reshape(-1, *full.shape[1:])
    s0, s1 = row_block(rays.n_rays, rank, world_size)
    return full[s0:s1]


class _GatherRays(torch.autograd.Function):
    """all-gather in forward; in backward every rank keeps the slice of the (rank-identical) full-frame gradient
    that belongs to its own rays."""

    @staticmethod
    def forward(ctx, t, rays):
        ctx.rays = rays
        out = gather_rays(t, rays)
        return out.reshape(-1, *t.shape[1:])

    @staticmethod
    def backward(ctx, g):
        return local_slice(g, ctx.rays).contiguous(), None


def gather_rays_autograd(t, rays: RaySet):
    """``gather_rays`` under autograd, returned flat: (n_rays_full, ...)."""
    if world()[1] == 1:
        return t
    return _GatherRays.apply(t, rays)


class _GradSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if g.is_cuda and dist.get_backend() == 'gloo':   # single-GPU test rigs only
            h = g.cpu()
            dist.all_reduce(h)
            return h.to(g.device)
        dist.all_reduce(g)           # RCCL: one sum of dL/d(volume) over xGMI per iteration
        return g


def replicate_grad_sum(t):
    """Mark a tensor that is REPLICATED on every rank and consumed by rank-local shards of the work: identity in
    forward, all-reduce(sum) of its gradient in backward, so that each rank ends up with the gradient of the whole
    (unsharded) loss.  For the ray-sharded head this is dL/d(sdf volume) and dL/d(feature volume)."""
    if t is None or world()[1] == 1:
        return t
    return _GradSum.apply(t)


class RayShard:
    """What a loss needs to know about a ray-sharded head output: the UNSHARDED lattice, this rank's shard of it and
    the pixel coordinates of the local rays (one camera's rows, shared by all cameras)."""

    # head outputs that hold only this rank's samples under ray sharding (the head also lists the shard itself under
    # outputs['ray_shard']: losses read it from there — a python attribute on a tensor does not survive .float() / .contiguous())
    local_keys = ('weights', 'ts', 'deltas', 'ray_indices', 'sample_sdf', 'eik_grad')

    def __init__(self, full: RaySet, local: RaySet, pix_local):
        self.full, self.local, self.pix_local = full, local, pix_local
        self.rank, self.world_size = world()

    @property
    def rays_per_cam_local(self):
        return self.local.nx * self.local.ny

    @property
    def rays_per_cam_full(self):
        return self.full.nx * self.full.ny


class LocalRows(list):
    """Per-camera list of per-sample tensors that hold only THIS RANK's rows of the lattice (ray-sharded training).
    Losses that declare ``supports_ray_shard`` reduce them to per-ray terms locally and gather those (``.shard``)."""

    def __init__(self, items, shard: RayShard):
        super().__init__(items)
        self.shard = shard


def tag_local(t, shard: RayShard):
    """Mark a tensor as holding only this rank's samples (e.g. eik_grad)."""
    t._so_shard = shard
    return t


def shard_of(x):
    """The RayShard of a head output (LocalRows / tagged tensor), or None."""
    if isinstance(x, LocalRows):
        return x.shard
    return getattr(x, '_so_shard', None)


def global_value_local_grad(local):
    """A per-rank partial sum whose parts add up to the loss term over ALL ranks: forward value = the global sum (what
    gets logged / compared), backward = the local part only (each rank back-propagates its own samples; the replicated
    volume's gradient is summed over the ranks by ``replicate_grad_sum``)."""
    if world()[1] == 1:
        return local
    tot = local.detach().clone()
    if tot.is_cuda and dist.get_backend() == 'gloo':
        h = tot.cpu()
        dist.all_reduce(h)
        tot = h.to(local.device)
    else:
        dist.all_reduce(tot)
    return local + (tot - local.detach())
