"""world_size-2 gloo test (CPU) of the ray-sharded multi-GPU path: shard -> render (the CPU
oracle stands in for the kernel here; there is no GPU in this container) -> gather /
all-reduce -> identical to the unsharded result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from selfocc_amd import synthetic as sy
from selfocc_amd.dist import row_block, shard_rays, gather_rays, all_reduce_mean


def test_row_block_partition():
    for n in (1, 7, 450, 451):
        for ws in (1, 2, 3, 8):
            blocks = [row_block(n, r, ws) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, ws, port, explicit, ret):
    import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        torch.set_num_threads(2)
        vol = sy.make_volume("cfg1", n_rgb=3, n_sem=0, seed=2)
        rays = sy.make_rays("cfg1", seed=2)
        rays.img2lidar = rays.img2lidar.repeat(2, 1, 1)            # 2 cameras, ny = 25 rows (uneven split)
        if explicit:
            rays = sy.explicit_rays(rays)
        cfg = sy.make_render_config("cfg1")
        full = oracle.render_fwd(vol, rays, cfg)
        mine = shard_rays(rays)
        out = oracle.render_fwd(vol, mine, cfg)
        depth = gather_rays(out['depth'], rays).reshape(-1)
        rgb = gather_rays(out['rgb'], rays).reshape(-1, 3)
        loss = all_reduce_mean(out['depth'].mean(), mine.n_rays)
        if explicit:
            same = torch.equal(depth, full['depth']) and torch.equal(rgb, full['rgb'])
        else:  # the shard's lattice offset oy + r0 * sy rounds differently from (iy + r0) * sy + oy
            same = torch.allclose(depth, full['depth'], rtol=1e-4, atol=1e-5) and \
                torch.allclose(rgb, full['rgb'], rtol=1e-4, atol=1e-5)
        ok = same and torch.allclose(loss, full['depth'].mean(), rtol=1e-5)
        ret[rank] = bool(ok) and mine.n_rays < rays.n_rays
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("explicit", [False, True])
def test_ray_sharding_world2_gloo(explicit):
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, explicit, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(ws)), dict(ret)


def _worker_autograd(rank, ws, port, ret):
    """Sharded autograd == unsharded autograd: a replicated 'volume', rays split by rows, a lattice-shaped loss
    on the gathered per-ray outputs.  A toy differentiable 'render' stands in for the kernel."""
    from selfocc_amd.dist import gather_rays_autograd, replicate_grad_sum, local_slice
    from selfocc_amd.render import RaySet
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        g = torch.Generator().manual_seed(0)
        vol = torch.randn(16, generator=g, dtype=torch.float64)
        n_cams, ny, nx = 2, 7, 5                                     # 7 rows: uneven split
        full = RaySet(img2lidar=torch.eye(4)[None].repeat(n_cams, 1, 1), nx=nx, ny=ny, sx=1.0, sy=1.0)
        feats = torch.randn(n_cams * ny * nx, 16, generator=g, dtype=torch.float64)   # per-ray "geometry"
        target = torch.randn(n_cams, ny, nx, generator=g, dtype=torch.float64)

        def render(v, f):                                            # per ray: a nonlinear function of the volume
            return torch.tanh(f @ v), torch.sigmoid(f * v[None])     # (n,), (n, 16) "per-sample"

        def loss_fn(depth, samples):                                 # needs the whole lattice (neighbour differences)
            d = depth.reshape(n_cams, ny, nx)
            return ((d - target) ** 2).mean() + (d[:, 1:] - d[:, :-1]).abs().mean() + samples.pow(3).mean()

        v0 = vol.clone().requires_grad_(True)
        loss_fn(*render(v0, feats)).backward()
        v1 = vol.clone().requires_grad_(True)
        mine = local_slice(feats, full)
        assert mine.shape[0] < feats.shape[0]
        d, s = render(replicate_grad_sum(v1), mine)
        loss = loss_fn(gather_rays_autograd(d, full), gather_rays_autograd(s, full))
        loss.backward()
        ret[rank] = bool(torch.allclose(v1.grad, v0.grad, rtol=1e-12, atol=1e-14))
    finally:
        dist.destroy_process_group()


def test_sharded_autograd_equals_unsharded_world2_gloo():
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_autograd, args=(r, ws, port, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(ws)), dict(ret)


def _worker_local_terms(rank, ws, port, ret):
    """Sharded loss terms: a per-rank partial sum reports the GLOBAL value and back-propagates the local part only; a
    loss class that does not know how to reduce local per-sample inputs refuses them."""
    from selfocc_amd.dist import global_value_local_grad, LocalRows, RayShard, tag_local
    from selfocc_amd.render import RaySet
    from selfocc_amd.loss import EikonalLoss, SecondGradLoss
    from selfocc_amd.loss.base import BaseLoss
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        ok = True
        x = torch.tensor([1.0 + rank, 2.0], requires_grad=True)
        v = global_value_local_grad((x ** 2).sum())
        v.backward()
        ok &= abs(v.item() - ((1 + 4) + (4 + 4))) < 1e-6 and torch.allclose(x.grad, 2 * x.detach())
        # EikonalLoss on this rank's samples == its share of the global mean (value: the global mean on every rank)
        full = RaySet(img2lidar=torch.eye(4)[None].repeat(2, 1, 1), nx=5, ny=7, sx=1.0, sy=1.0)
        from selfocc_amd.dist import shard_rays, row_block
        local = shard_rays(full)
        shard = RayShard(full, local, torch.zeros(local.nx * local.ny, 2))
        S = 3
        g = torch.Generator().manual_seed(0)
        grads_all = torch.randn(2, 7, 5, S, 3, generator=g, dtype=torch.float64)
        r0, r1 = row_block(7, rank, ws)
        mine = grads_all[:, r0:r1].reshape(-1, 3).clone().requires_grad_(True)
        val = EikonalLoss(weight=1.0)(dict(eik_grad=tag_local(mine, shard)))
        want = ((grads_all.reshape(-1, 3).norm(2, dim=-1) - 1) ** 2).mean()
        ok &= bool(torch.allclose(val.detach(), want, rtol=1e-12))
        val.backward()
        ref = grads_all.reshape(-1, 3).clone().requires_grad_(True)
        ((ref.norm(2, dim=-1) - 1) ** 2).mean().backward()
        ok &= bool(torch.allclose(mine.grad, ref.grad.reshape(2, 7, 5, S, 3)[:, r0:r1].reshape(-1, 3), rtol=1e-12))

        class Naive(BaseLoss):          # a loss that would silently average local rows as if they were the frame
            def __init__(self):
                super().__init__(1.0, {'ts': 'ts'})
                self.loss_func = lambda ts: torch.stack(list(ts)).mean()
        try:
            Naive()(dict(ts=LocalRows([torch.ones(4)], shard)))
            ok = False
        except NotImplementedError:
            pass
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_sharded_loss_terms_world2_gloo():
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_local_terms, args=(r, ws, port, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(ws)), dict(ret)


# ---- encoder row sharding: the collectives and the bookkeeping on CPU tensors (the encoder's kernels need a GPU: the
# two-rank comparison with the real modules is tests/test_dist_gpu.py::test_row_sharded_encoder_equals_unsharded_world2)
def test_plane_row_shard_bookkeeping():
    from selfocc_amd.dist import PlaneRowShard
    sizes = [13, 5, 7]
    full = torch.arange(sum(sizes) * 2, dtype=torch.float32).reshape(1, sum(sizes), 2)
    for ws in (2, 3, 5):
        shards = [PlaneRowShard(sizes, r, ws) for r in range(ws)]
        assert sum(s.n_local for s in shards) == sum(sizes)
        assert all(max(s.local_sizes[i] for s in shards) - min(s.local_sizes[i] for s in shards) <= 1 for i in range(3))
        # pad every rank's local rows to max_local, stack, index: the concatenated planes come back in order
        pad = torch.zeros(ws * shards[0].max_local, 2)
        for r, s in enumerate(shards):
            pad[r * s.max_local:r * s.max_local + s.n_local] = s.take(full, 1)[0]
        assert torch.equal(pad.index_select(0, shards[0].gather_index('cpu'))[None], full)
        planes = list(torch.split(full, sizes, 1))
        assert all(torch.equal(s.take(planes, 1), s.take(full, 1)) for s in shards)
    with pytest.raises(AssertionError):
        PlaneRowShard([4, 1, 4], 0, 2)


def _worker_rows(rank, ws, port, ret):
    from selfocc_amd.dist import PlaneRowShard, gather_plane_rows, group_grad_sum, replicate_grad_sum
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        g = torch.Generator().manual_seed(0)
        sizes, C = [11, 4, 6], 3
        x = torch.randn(1, sum(sizes), C, generator=g)                  # replicated input planes
        w1, w2 = torch.randn(C, C, generator=g), torch.randn(C, C, generator=g)
        coef = torch.randn(1, sum(sizes), C, generator=g)

        def layers(xf, a, b, shard):
            """two 'layers': rows mix with the column means of ALL rows (the value path) through a shared weight"""
            for li, w in enumerate((a, b)):
                if shard is None:
                    xf = torch.tanh(xf @ w) + xf.mean(1, keepdim=True) @ w
                else:
                    (wl,) = group_grad_sum([w])
                    loc = torch.tanh(shard.take(xf, 1) @ wl) + xf.mean(1, keepdim=True) @ wl
                    xf = gather_plane_rows(loc, shard, reduce_grad=li < 1)
            return xf

        xa, a0, b0 = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
        ref = layers(xa, a0, b0, None)
        (ref * coef).sum().backward()
        xb, a1, b1 = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
        got = layers(replicate_grad_sum(xb), a1, b1, PlaneRowShard(sizes))
        (got * coef).sum().backward()                                     # the same (replicated) loss on every rank
        ok = torch.allclose(got, ref, atol=1e-6)
        for p, q in ((xa, xb), (a0, a1), (b0, b1)):
            ok = ok and torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6)
        # default: SYNCHRONOUS all-reduces — what autograd hands on is already reduced (round 6; see _worker_grad_readers)
        from selfocc_amd import dist as sdist
        ok = ok and sdist.OVERLAP_STATS == {'deferred': 0, 'synchronous': 2} and not sdist._PENDING
        # opted in: launched async, waited for and written into .grad by the end-of-backward callback — also when a
        # parameter already holds a gradient (the accumulation happens after the wait)
        sdist.enable_overlap(True)
        got2 = layers(replicate_grad_sum(xb), a1, b1, PlaneRowShard(sizes))
        (got2 * coef).sum().backward()
        sdist.enable_overlap(False)
        ok = ok and sdist.OVERLAP_STATS == {'deferred': 2, 'synchronous': 2} and not sdist._PENDING
        ok = ok and torch.allclose(a1.grad, 2 * a0.grad, rtol=1e-5, atol=1e-6) and torch.allclose(b1.grad, 2 * b0.grad, rtol=1e-5, atol=1e-6)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_row_sharded_layers_equal_unsharded_world2_gloo():
    """gather_plane_rows / group_grad_sum / replicate_grad_sum: a two-layer stand-in whose rows read ALL rows (like the
    cross-view self-attention's value) gives the unsharded outputs and gradients on both ranks."""
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rows, args=(r, ws, port, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) is True for r in range(ws)), dict(ret)


def _worker_grad_readers(rank, ws, port, ret):
    """Everything that can read a ``group_grad_sum`` gradient before / while it is reduced (round-5 review weak #5 + advisor):
    a post-accumulate hook, a DistributedDataParallel wrapper (train.py:85-91 always wraps), gradient accumulation
    (train.py:236-242), forward-forward-backward-backward on the same parameters, a bfloat16 parameter, and a backward
    that raises after queueing work — with the overlap off (default) and opted in."""
    from selfocc_amd import dist as sdist
    from selfocc_amd.dist import PlaneRowShard, gather_plane_rows, group_grad_sum, replicate_grad_sum
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        g = torch.Generator().manual_seed(1)
        sizes, C = [9, 4, 5], 3
        x = torch.randn(1, sum(sizes), C, generator=g)
        W = [torch.randn(C, C, generator=g) for _ in range(2)]
        coef = torch.randn(1, sum(sizes), C, generator=g)
        shard = PlaneRowShard(sizes)

        class Net(torch.nn.Module):
            def __init__(self, dtype=torch.float32):
                super().__init__()
                self.w = torch.nn.ParameterList([torch.nn.Parameter(w.clone().to(dtype)) for w in W])

            def forward(self, xf, shard=None):
                for li, w in enumerate(self.w):
                    if shard is None:
                        xf = torch.tanh(xf @ w.float()) + xf.mean(1, keepdim=True) @ w.float()
                    else:
                        (wl,) = group_grad_sum([w])
                        loc = torch.tanh(shard.take(xf, 1) @ wl.float()) + xf.mean(1, keepdim=True) @ wl.float()
                        xf = gather_plane_rows(loc, shard, reduce_grad=li < 1)
                return xf

        def ref_grads(dtype=torch.float32, times=1):
            net = Net(dtype)
            for _ in range(times):
                (net(x) * coef).sum().backward()
            return [p.grad.clone() for p in net.parameters()]

        def close(a, b, tol=1e-5):
            return all(torch.allclose(p.float(), q.float(), rtol=tol, atol=tol) for p, q in zip(a, b))

        ok, want, want2 = {}, ref_grads(), ref_grads(times=2)
        for overlap in (False, True):
            sdist.enable_overlap(overlap)
            tag = 'overlap' if overlap else 'sync'
            # (a) a post-accumulate hook sees the REDUCED gradient (with the overlap on, the hook forces the synchronous route)
            net, seen = Net(), []
            for p in net.parameters():
                p.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.clone()))
            before = dict(sdist.OVERLAP_STATS)
            (net(x, shard) * coef).sum().backward()
            ok[f'hook_{tag}'] = len(seen) == 2 and close(seen[::-1], want) and close([p.grad for p in net.parameters()], want) \
                and sdist.OVERLAP_STATS['deferred'] == before['deferred']
            # (c) gradient accumulation over two backward passes
            net = Net()
            for _ in range(2):
                (net(x, shard) * coef).sum().backward()
            ok[f'accum_{tag}'] = close([p.grad for p in net.parameters()], want2)
            # forward, forward, backward, backward on the same parameters (advisor case 1)
            net = Net()
            l1, l2 = (net(x, shard) * coef).sum(), (net(x, shard) * coef).sum()
            l1.backward()
            l2.backward()
            ok[f'ffbb_{tag}'] = close([p.grad for p in net.parameters()], want2)
            # one backward over two uses of the parameters
            net = Net()
            ((net(x, shard) * coef).sum() + (net(x, shard) * coef).sum()).backward()
            ok[f'twice_{tag}'] = close([p.grad for p in net.parameters()], want2)
            # a bfloat16 parameter (advisor case 2): the cast happens after the reduction
            net = Net(torch.bfloat16)
            (net(x, shard) * coef).sum().backward()
            ok[f'bf16_{tag}'] = all(p.grad.dtype == torch.bfloat16 for p in net.parameters()) and \
                close([p.grad for p in net.parameters()], ref_grads(torch.bfloat16), 2e-2)
            # a backward that raises after the all-reduce was queued: the next forward waits for the stale work, drops it,
            # and the callback is registered again per backward pass (advisor, low)
            net = Net()

            class Boom(torch.autograd.Function):
                @staticmethod
                def forward(ctx, t):
                    return t.view_as(t)

                @staticmethod
                def backward(ctx, g):
                    raise RuntimeError("boom")
            # the failing node sits on the INPUT: it runs after the layers' group sums were queued
            lossb = (net(Boom.apply(x.clone().requires_grad_(True)), shard) * coef).sum()
            try:
                lossb.backward()
                raised = False
            except RuntimeError:
                raised = True
            ok[f'raise_left_work_{tag}'] = (len(sdist._PENDING) > 0) == overlap
            net.zero_grad(set_to_none=True)
            (net(x, shard) * coef).sum().backward()
            # whatever the failed pass left behind was waited for and discarded: the good pass's gradient is exactly one
            # pass's worth, and nothing stays pending
            ok[f'raise_{tag}'] = raised and not sdist._PENDING and sdist._CALLBACK_QUEUED[0] == -1 and \
                close([p.grad for p in net.parameters()], want)
        sdist.enable_overlap(False)
        # (b) DistributedDataParallel around the row-sharded stand-in: DDP's bucket hook reads the gradient at accumulate
        # time and averages N identical, already reduced gradients == the unsharded gradient
        from torch.nn.parallel import DistributedDataParallel as DDP
        ddp = DDP(Net())
        for it in range(2):                                   # second iteration: DDP's steady-state bucket views
            ddp.zero_grad(set_to_none=True)
            (ddp(x, shard) * coef).sum().backward()
            ok[f'ddp_iter{it}'] = close([p.grad for p in ddp.module.parameters()], want)
        # ... and with accumulation under no_sync (train.py:236-242 accumulates over grad_accumulation passes)
        ddp.zero_grad(set_to_none=True)
        with ddp.no_sync():
            (ddp(x, shard) * coef).sum().backward()
        (ddp(x, shard) * coef).sum().backward()
        ok['ddp_accum'] = close([p.grad for p in ddp.module.parameters()], want2)
        ret[rank] = {k: bool(v) for k, v in ok.items()}
    finally:
        dist.destroy_process_group()


def test_group_grad_sum_is_safe_under_hooks_ddp_accumulation_world2_gloo():
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_grad_readers, args=(r, ws, port, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for r in range(ws):
        bad = [k for k, v in (ret.get(r) or {'missing': False}).items() if not v]
        assert not bad, (r, bad)
