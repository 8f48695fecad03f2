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
