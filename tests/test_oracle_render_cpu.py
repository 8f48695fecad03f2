"""CPU: pin the C oracle (oracle/oracle_render.c) against the torch ops the reference
calls (oracle/torch_port.py) and against torch primitives."""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_port as tp
from selfocc_amd import abi, synthetic as sy


def test_canonical_expf_close_to_libm():
    xs = np.concatenate([np.linspace(-87, 88, 4001), np.random.RandomState(0).uniform(-20, 20, 2000)])
    got = np.array([oracle.expf(float(x)) for x in xs.astype(np.float32)], dtype=np.float64)
    ref = np.exp(xs.astype(np.float32).astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 2.5e-7  # <= ~2 ulp


@pytest.mark.parametrize("n", [1, 2, 3, 7, 32, 100, 127, 128, 255, 256])
def test_linspace_matches_torch(n):
    ref = torch.linspace(0.0, 1.0, n + 1)
    got = torch.tensor([oracle.linspace01(j, n) for j in range(n + 1)])
    assert torch.equal(ref, got)


@pytest.mark.parametrize("name", ["cfg1", "cfg5"])
def test_trilinear_bit_exact_vs_grid_sample(name):
    """The lookup the reference performs (bev_nerf.py:103-113) == oracle trilinear, bit for bit."""
    vol = sy.make_volume(name, seed=1)
    m = vol.mapping
    g = torch.Generator().manual_seed(2)
    lo = torch.tensor(sy.CONFIGS[name]["aabb"][:3]); hi = torch.tensor(sy.CONFIGS[name]["aabb"][3:])
    xyz = lo + (hi - lo) * torch.rand(20000, 3, generator=g)
    # include exact boundary / grid-point hits
    xyz[:8] = torch.stack([torch.stack([a, b, c]) for a in (lo[0], hi[0]) for b in (lo[1], hi[1]) for c in (lo[2], hi[2])])
    ref = tp.field_lookup(m, vol.sdf[None, None], xyz)[:, 0]
    got, grad = oracle.field_sdf(m, vol.sdf, xyz)
    assert torch.equal(ref, got)
    # gradient vs autograd through grid_sample
    p = xyz.clone().requires_grad_(True)
    tp.field_lookup(m, vol.sdf[None, None], p)[:, 0].sum().backward()
    # (the 8 box-corner points sit on sign()/abs() kinks where autograd returns 0)
    assert torch.allclose(p.grad[8:], grad[8:], rtol=1e-5, atol=1e-5)


def _port_inputs(name, n_rgb, n_sem, seed=0):
    vol = sy.make_volume(name, n_rgb=n_rgb, n_sem=n_sem, seed=seed)
    rays = sy.make_rays(name, seed)
    return vol, rays


@pytest.mark.parametrize("n_rgb,n_sem,sample_pos", [(0, 0, 0), (3, 0, 1), (3, 5, 0)])
def test_c_oracle_vs_torch_port_cfg1(n_rgb, n_sem, sample_pos):
    vol, rays = _port_inputs("cfg1", n_rgb, n_sem)
    cfg = sy.make_render_config("cfg1", inv_s=20.0, sample_pos=sample_pos, bkgd_mode=abi.BKGD_CONST,
                                bkgd=(1.0, 1.0, 1.0))
    got = oracle.render_fwd(vol, rays, cfg, per_sample=True, want_grad_samples=True)
    ex = sy.explicit_rays(rays)
    ref = tp.render_port(vol.mapping, vol.to_reference_layout(), n_rgb, n_sem, ex.origins, ex.dirs,
                         ex.dir_norm, cfg, return_samples=True)
    # geometry + lookup agree to float rounding
    assert torch.allclose(got['fars'], ref['fars'], rtol=1e-5, atol=1e-5)
    assert torch.allclose(got['ts'], ref['ts'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(got['sdf'], ref['sdf'], rtol=1e-4, atol=2e-5)
    assert torch.allclose(got['grad'], ref['grad'], rtol=1e-3, atol=1e-4)
    # NeuS weights: the sigmoid difference cancels catastrophically below ~1e-5, so compare
    # absolutely (weights are <= 1) and the composites relatively where acc is not negligible
    assert torch.allclose(got['weights'], ref['weights'], rtol=2e-3, atol=2e-6)
    ok = ref['acc'] > 0.05
    assert ok.sum() > 100
    assert torch.allclose(got['acc'][ok], ref['acc'][ok], rtol=1e-4, atol=1e-5)
    assert torch.allclose(got['depth'][ok], ref['depth'][ok], rtol=1e-4, atol=1e-5)
    if n_rgb:
        assert torch.allclose(got['rgb'], ref['rgb'], rtol=1e-4, atol=2e-5)
    if n_sem:
        assert torch.allclose(got['sem'], ref['sem'], rtol=1e-4, atol=2e-5)
    same = (got['max_depth'] == ref['max_depth']).float().mean()
    assert same > 0.98 or torch.allclose(got['max_depth'], ref['max_depth'], rtol=1e-5, atol=1e-5)


def test_explicit_equals_pixel_grid_oracle():
    vol, rays = _port_inputs("cfg1", 3, 0)
    cfg = sy.make_render_config("cfg1")
    a = oracle.render_fwd(vol, rays, cfg)
    b = oracle.render_fwd(vol, sy.explicit_rays(rays), cfg)
    for k in a:
        assert torch.allclose(a[k], b[k], rtol=1e-4, atol=1e-5), k


def test_jitter_modes_oracle_vs_port():
    vol, rays = _port_inputs("cfg1", 0, 0)
    ex = sy.explicit_rays(rays)
    g = torch.Generator().manual_seed(5)
    for mode, shape in [(abi.JITTER_SINGLE, (ex.n_rays,)), (abi.JITTER_PER_BIN, (ex.n_rays, 33))]:
        cfg = sy.make_render_config("cfg1", jitter_mode=mode)
        t_rand = torch.rand(*shape, generator=g)
        got = oracle.render_fwd(vol, ex, cfg, per_sample=True, t_rand=t_rand)
        ref = tp.render_port(vol.mapping, vol.to_reference_layout(), 0, 0, ex.origins, ex.dirs, ex.dir_norm,
                             cfg, t_rand=t_rand, return_samples=True)
        assert torch.allclose(got['ts'], ref['ts'], rtol=1e-5, atol=1e-6)
        assert torch.allclose(got['deltas'], ref['deltas'], rtol=1e-4, atol=1e-6)


def test_compositing_invariants():
    vol, rays = _port_inputs("cfg1", 3, 5)
    cfg = sy.make_render_config("cfg1", inv_s=50.0)
    o = oracle.render_fwd(vol, rays, cfg, per_sample=True)
    assert (o['weights'] >= 0).all() and (o['acc'] <= 1.0 + 1e-4).all()
    assert (o['depth'] * 1.0 >= 0).all()
    assert torch.allclose(o['weights'].sum(-1), o['acc'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(o['sem'].sum(-1), o['acc'], rtol=1e-4, atol=1e-5)  # softmax sums to 1


def test_differentiable_port_matches_grid_sample_port():
    """trilinear_explicit (used for the backward parity tests) == F.grid_sample lookup."""
    vol, rays = _port_inputs("cfg1", 3, 5)
    ex = sy.explicit_rays(rays)
    cfg = sy.make_render_config("cfg1", inv_s=15.0)
    a = tp.render_port(vol.mapping, vol.to_reference_layout(), 3, 5, ex.origins, ex.dirs, ex.dir_norm, cfg,
                       return_samples=True)
    b = tp.render_port_differentiable(vol.mapping, vol.to_reference_layout()[0].double(), 3, 5, ex.origins.double(),
                                      ex.dirs.double(), ex.dir_norm.double(), cfg, torch.tensor(15.0, dtype=torch.float64))
    for k in ('sdf', 'grad', 'weights', 'acc', 'rgb', 'sem'):
        assert torch.allclose(a[k].double(), b[k], rtol=2e-3, atol=2e-5), k


def test_oracle_with_hand_filled_args_equals_marshalled_args():
    """oracle.render_fwd fills `so_render_args` through the product's marshal_render_args, so a marshalling bug would be
    invisible to a HIP-vs-oracle comparison (both sides read the same struct).  Here the struct is
    filled field by field from raw tensors and plain numbers, following include/selfocc_hip.h only — the C oracle must
    return the same bits either way; the torch port (explicit rays built here from the matrices, not by the product's
    helpers) pins the meaning of the lattice fields."""
    import ctypes as C
    n_rgb, n_sem = 3, 5
    vol = sy.make_volume("cfg1", n_rgb=n_rgb, n_sem=n_sem, seed=4)
    rays = sy.make_rays("cfg1", seed=4)
    cfg = sy.make_render_config("cfg1", inv_s=25.0, sample_pos=1, bkgd_mode=abi.BKGD_CONST, bkgd=(0.25, 0.5, 1.0),
                                clamp_rgb=True, exact=True)
    ref = oracle.render_fwd(vol, rays, cfg, per_sample=True, want_grad_samples=True)

    a = abi.SoRenderArgs()
    a.map = vol.mapping.to_abi()
    sdf, feat, cams = vol.sdf.contiguous(), vol.feat.contiguous(), rays.img2lidar.contiguous().float()
    a.sdf_vol, a.feat_vol = sdf.data_ptr(), feat.data_ptr()
    a.feat_dtype, a.feat_stride, a.n_rgb, a.n_sem = 0, feat.shape[3], n_rgb, n_sem          # SO_DTYPE_F32 = 0
    n_cams, ny, nx = cams.shape[0], rays.ny, rays.nx
    N, S = n_cams * ny * nx, 32
    a.ray_mode, a.n_rays = 1, N                                                              # SO_RAYS_PIXEL_GRID = 1
    a.img2lidar, a.n_cams, a.nx, a.ny = cams.data_ptr(), n_cams, nx, ny
    a.sx, a.sy, a.ox, a.oy = rays.sx, rays.sy, rays.ox, rays.oy
    for i, v in enumerate(sy.CONFIGS["cfg1"]["aabb"]):
        a.aabb[i] = v
    a.near_plane, a.n_samples, a.sample_pos, a.jitter_mode = 0.0, S, 1, 0                    # SO_SAMPLE_AT_MID, no jitter
    a.inv_s, a.bkgd_mode = 25.0, 1                                                           # SO_BKGD_CONST = 1
    a.bkgd[0], a.bkgd[1], a.bkgd[2] = 0.25, 0.5, 1.0
    a.flags = 1 | 2 | 4                                                                      # DEPTH_DIV_NORM | CLAMP_RGB | EXACT
    shapes = dict(depth=(N,), acc=(N,), max_depth=(N,), nears=(N,), fars=(N,), rgb=(N, 3), sem=(N, n_sem),
                  weights=(N, S), ts=(N, S), deltas=(N, S), sdf=(N, S), grad=(N, S, 3))
    out = {k: torch.empty(*s) for k, s in shapes.items()}
    for k, t in out.items():
        setattr(a, k, t.data_ptr())
    # the constants above are the header's (a drift between abi.py and these literals fails here, not silently)
    assert (abi.DTYPE_F32, abi.RAYS_PIXEL_GRID, abi.SAMPLE_AT_MID, abi.JITTER_NONE, abi.BKGD_CONST) == (0, 1, 1, 0, 1)
    assert (abi.FLAG_DEPTH_DIV_NORM, abi.FLAG_CLAMP_RGB, abi.FLAG_EXACT) == (1, 2, 4)
    fn = oracle.lib().oracle_render_fwd
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(abi.SoRenderArgs)]
    assert fn(a) == 0
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    # meaning of the lattice fields: pixel (ix * sx + ox, iy * sy + oy) of camera c, row-major, cameras outermost
    xs = torch.arange(nx, dtype=torch.float32) * rays.sx + rays.ox
    ys = torch.arange(ny, dtype=torch.float32) * rays.sy + rays.oy
    pix = torch.stack([xs[None].expand(ny, -1), ys[:, None].expand(-1, nx), torch.ones(ny, nx)], -1).reshape(-1, 3)
    d = torch.einsum('cij,pj->cpi', cams[:, :3, :3], pix).reshape(-1, 3)
    o = cams[:, None, :3, 3].expand(-1, ny * nx, -1).reshape(-1, 3)
    dn = d.norm(dim=-1)
    port = tp.render_port(vol.mapping, vol.to_reference_layout(), n_rgb, n_sem, o, d / dn[:, None], dn, cfg)
    ok = port['acc'] > 0.05
    assert torch.allclose(out['fars'], port['fars'], rtol=1e-5, atol=1e-5)
    assert torch.allclose(out['depth'][ok], port['depth'][ok], rtol=1e-4, atol=1e-5)
    assert torch.allclose(out['rgb'], port['rgb'], rtol=1e-4, atol=2e-5)
