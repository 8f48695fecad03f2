"""Multi-GPU layer of the hot path: one process per GPU, ``torch.distributed`` with backend
"nccl" (= RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).

Rays are independent units and the (small) field volume is replicated, so the path shards
with NO data-path collective:
  * ``shard_rays``     contiguous row blocks of every camera's ray lattice per rank (keeps the
                       image locality the 8x8 wavefront tiles and the SSIM windows rely on);
  * eval:              every rank scores its shard, the integer / float metric sums are
                       all-reduced once per epoch (MeanIoU._after_epoch), or the rendered maps
                       are ``gather_rays``-ed;
  * train:             each rank back-propagates its shard's loss; the gradient of the
                       replicated field w.r.t. the tri-planes / MLP is summed by the usual DDP
                       bucketed all-reduce, ``all_reduce_mean`` reports the global loss.
The reference itself only has frame-per-GPU DDP (train.py:86-91).
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
        dist.all_gather(bufs, pad.contiguous())
        return torch.cat([b[:, :(r1 - r0)] for b, (r0, r1) in zip(bufs, rows)], dim=1)
    sizes = [row_block(rays.n_rays, r, ws) for r in range(ws)]
    mx = max(b - a for a, b in sizes)
    pad = t.new_zeros(mx, *t.shape[1:])
    pad[:t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad.contiguous())
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
