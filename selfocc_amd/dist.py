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
import os

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


# ---- encoder row sharding ("shard queries across ranks, values replicated, all-gather the updated
# planes per layer") ------------------------------------------------------------------------------------------------------
def _all_reduce_(t):
    """in-place sum over the ranks (gloo cannot reduce CUDA tensors: staged through the host on the test rigs)"""
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


class PlaneRowShard:
    """Row blocks of the TPV / BEV planes per rank: EVERY plane is cut into world_size contiguous blocks (a zh / wz row costs
    six times the sampling points of an hw row at the shipped pillar sizes, so cutting the concatenated tensor instead would
    leave the last ranks with most of the work).  ``local`` rows of a rank = its block of plane 0, then of plane 1, ..."""

    def __init__(self, sizes, rank=None, world_size=None):
        if rank is None:
            rank, world_size = world()
        assert min(sizes) >= world_size, f"row sharding needs >= {world_size} rows in every plane, got {sizes}"
        self.sizes, self.rank, self.world_size = list(sizes), rank, world_size
        self.blocks = [[row_block(n, r, world_size) for n in sizes] for r in range(world_size)]     # [rank][plane] = (a, b)
        self.local = self.blocks[rank]
        self.local_sizes = [b - a for a, b in self.local]
        self.n_local = sum(self.local_sizes)
        self.max_local = max(sum(b - a for a, b in blk) for blk in self.blocks)
        self._index = {}

    def take(self, x, dim=1):
        """local rows of a tensor whose ``dim`` runs over the concatenated planes, or of a list of per-plane tensors"""
        if torch.is_tensor(x):
            off, parts = 0, []
            for n, (a, b) in zip(self.sizes, self.local):
                parts.append(x.narrow(dim, off + a, b - a))
                off += n
            return torch.cat(parts, dim)
        return torch.cat([t.narrow(dim, a, b - a) for t, (a, b) in zip(x, self.local)], dim)

    def take_plane(self, t, i, dim):
        a, b = self.local[i]
        return t.narrow(dim, a, b - a)

    def gather_index(self, device):
        """row of the (world_size * max_local) padded all-gather buffer that holds row j of the concatenated planes"""
        idx = self._index.get(str(device))
        if idx is None:
            parts = []
            for i in range(len(self.sizes)):
                for r in range(self.world_size):
                    off = sum(b - a for a, b in self.blocks[r][:i])
                    a, b = self.blocks[r][i]
                    parts.append(torch.arange(b - a) + (r * self.max_local + off))
            idx = self._index[str(device)] = torch.cat(parts).to(device)
        return idx


def _gather_plane_rows(loc, shard):
    """(1, n_local, C) of every rank -> (1, N, C) in plane order"""
    C = loc.shape[-1]
    pad = loc.new_zeros(shard.max_local, C)
    pad[:shard.n_local] = loc.reshape(-1, C)
    bufs = [torch.empty_like(pad) for _ in range(shard.world_size)]
    _all_gather(bufs, pad)
    return torch.cat(bufs, 0).index_select(0, shard.gather_index(loc.device)).unsqueeze(0)


class _GatherPlaneRows(torch.autograd.Function):
    """all-gather of the ranks' row blocks in forward.  Backward: the gathered tensor is consumed by EVERY rank (as the
    next layer's local rows and as its self-attention value), each contributing a part of its gradient, so the parts are
    summed over the ranks first (``reduce_grad``; not for the encoder's final output, whose gradient arrives complete
    and identical on every rank from the replicated field volume) and each rank keeps the rows it produced."""

    @staticmethod
    def forward(ctx, loc, shard, reduce_grad):
        ctx.shard, ctx.reduce_grad = shard, reduce_grad
        return _gather_plane_rows(loc, shard)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if ctx.reduce_grad:
            g = _all_reduce_(g.clone())
        return ctx.shard.take(g, 1).contiguous(), None, None


def gather_plane_rows(loc, shard, reduce_grad=True):
    if torch.is_grad_enabled() and loc.requires_grad:
        return _GatherPlaneRows.apply(loc, shard, reduce_grad)
    return _gather_plane_rows(loc, shard)


_PENDING = []          # (work, flat, params) of the all-reduces launched during the running backward pass
_CALLBACK_QUEUED = [-1]     # id of the backward pass (graph task) whose end-of-backward callback is registered
_OVERLAP = [os.environ.get('SELFOCC_DIST_OVERLAP', '0') == '1']
OVERLAP_STATS = {'deferred': 0, 'synchronous': 0}


def enable_overlap(on=True):
    """Opt in to (out of) the deferred parameter-gradient all-reduce of ``group_grad_sum``.  OFF by default: the deferred
    route hands the gradients to ``.grad`` itself at the end of ``backward()`` instead of through autograd's AccumulateGrad,
    so a ``DistributedDataParallel`` wrapper (the reference always wraps the model when distributed, train.py:85-91), a
    ``register_post_accumulate_grad_hook`` or ``torch.autograd.grad(inputs=parameters)`` would not see them.  The caller
    that owns the training loop and knows none of those is in use switches it on (SELFOCC_DIST_OVERLAP=1 does the same)."""
    _OVERLAP[0] = bool(on)
    return _OVERLAP[0]


def _flush_pending(apply=True):
    """end-of-backward callback: wait for every deferred all-reduce (a stream dependency on RCCL, no host block), THEN cast
    and add the reduced pieces into ``.grad`` on the compute stream — nothing ever reads a buffer that is still being
    reduced, whatever the parameter's dtype, whether a ``.grad`` already exists, and however many backward passes
    contributed to this flush.  ``apply=False``: the work of a backward pass that raised (the engine dropped its callbacks)
    is waited for and DISCARDED, like the gradients autograd itself had not accumulated yet."""
    _CALLBACK_QUEUED[0] = -1
    pending, _PENDING[:] = list(_PENDING), []
    for work, flat, params in pending:
        work.wait()
        if not apply:
            continue
        off = 0
        for p in params:
            n = p.numel()
            if p.requires_grad:
                g = flat[off:off + n].view(p.shape).to(p.dtype)           # after the wait: a window into reduced memory
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
            off += n


def _readers_attached(ts):
    """python-visible hooks that would read a gradient at accumulate time"""
    return any(getattr(t, '_post_accumulate_grad_hooks', None) or getattr(t, '_backward_hooks', None) for t in ts)


class _GroupGradSum(torch.autograd.Function):
    """Identity on a group of (parameter) tensors; backward = ONE coalesced all-reduce(sum) of all their gradients: under
    row sharding a rank's parameter gradients cover its own rows only.

    Default (round 6): SYNCHRONOUS — the gradients autograd hands on (to AccumulateGrad, its hooks, a DDP reducer) are
    already reduced.  Round 5 launched the all-reduce ``async_op=True`` and returned views of the buffer being reduced;
    ``.grad`` came out right only because AccumulateGrad aliased the view, and anything reading at accumulate time — a
    post-accumulate hook, DDP's bucket hook, a second backward adding into an existing ``.grad``, a bf16 cast — saw the
    rank-local partial (reproduced on two gloo ranks by the round-5 review; ``tests/test_dist_cpu.py`` now holds those cases).

    Opt-in overlap (``enable_overlap()``): the all-reduce is launched async on the collective stream beside the previous
    layers' backward kernels (59 MB per iteration at the shipped size), autograd receives NO gradient for the group, and the
    end-of-backward callback ``_flush_pending`` waits and then writes / accumulates ``.grad`` itself.  Decided at BACKWARD
    time; refused (synchronous) when a python hook is attached to a tensor of the group, when the tensors are not leaves,
    or when the backend cannot reduce device memory (the gloo-on-CUDA test rigs).
    The all-gather of a layer's rows and the all-reduce of the gathered planes' gradients (``gather_plane_rows``) are
    always synchronous: the next layer's first op / the previous layer's backward consume them immediately."""

    @staticmethod
    def forward(ctx, *ts):
        if _PENDING and torch._C._current_graph_task_id() == -1:
            _flush_pending(apply=False)     # left behind by a backward pass that raised after queueing work
        ctx.params = ts
        return tuple(t.view_as(t) for t in ts)

    @staticmethod
    def backward(ctx, *gs):
        ts = ctx.params
        ref = next(g for g in gs if g is not None)
        parts = [(g if g is not None else torch.zeros_like(t)).reshape(-1).float() for g, t in zip(gs, ts)]
        flat = torch.cat(parts) if len(parts) > 1 else parts[0].clone()
        staged = flat.is_cuda and dist.get_backend() == 'gloo'          # test rigs: gloo cannot reduce device memory
        defer = _OVERLAP[0] and not staged and all(t.is_leaf for t in ts) and not _readers_attached(ts)
        if defer:
            work = dist.all_reduce(flat, async_op=True)
            task = torch._C._current_graph_task_id()
            if _CALLBACK_QUEUED[0] != task:      # registered per backward pass, not "while the list is empty"
                if _PENDING:
                    _flush_pending(apply=False)  # a previous pass raised: its callback never ran
                torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
                _CALLBACK_QUEUED[0] = task
            _PENDING.append((work, flat, ts))
            OVERLAP_STATS['deferred'] += 1
            return tuple(None for _ in ts)
        _all_reduce_(flat)
        OVERLAP_STATS['synchronous'] += 1
        out, off = [], 0
        for g, t in zip(gs, ts):
            n = t.numel()
            out.append(flat[off:off + n].view(t.shape).to(ref.dtype if g is None else g.dtype))
            off += n
        return tuple(out)


def group_grad_sum(tensors):
    """``tensors`` unchanged in forward; their gradients summed over the ranks in one all-reduce in backward."""
    tensors = list(tensors)
    if world()[1] == 1 or not torch.is_grad_enabled() or not any(t.requires_grad for t in tensors):
        return tensors
    return list(_GroupGradSum.apply(*tensors))
