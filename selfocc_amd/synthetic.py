"""Seeded synthetic scenes with the shapes of BASELINE.json's configs.
No dataset or checkpoint exists in the build environment, so bench.py and the tests
drive the hot path with these: an SDF volume that is the signed distance to a union
of random boxes plus a ground plane (rays terminate like in a real scene), random
colour / semantic logits, and a ring of pinhole cameras at the centre of the box.
"""
import math

import numpy as np
import torch

from .mapping import GridMeterMapping
from .render import SDFVolume, RaySet, RenderConfig

CONFIGS = {
    # cfg1: single cam 64x64 image, 32x32x4 grid, 1k rays x 32 samples (CPU plumbing case)
    "cfg1": dict(mapping=dict(nonlinear_mode='linear', h_size=[31, 0], h_range=[12.8, 0], h_half=True,
                              w_size=[31, 0], w_range=[12.8, 0], w_half=True, d_size=[3, 0],
                              d_range=[-1.0, 2.0, 2.0]),
                 aabb=(0.0, 0.0, -1.0, 12.8, 12.8, 2.0), n_cams=1, img=(64, 64), rays=(25, 40),
                 n_samples=32, focal=60.0, cam_z=0.5),
    # cfg2: nuScenes-like 6 cams, 450x800 ray lattice on 900x1600 images, 200x200x16, 128 samples
    "cfg2": dict(mapping=dict(nonlinear_mode='linear', h_size=[199, 0], h_range=[80.0, 0], h_half=True,
                              w_size=[199, 0], w_range=[80.0, 0], w_half=True, d_size=[15, 0],
                              d_range=[-1.0, 5.4, 5.4]),
                 aabb=(0.0, 0.0, -1.0, 80.0, 80.0, 5.4), n_cams=6, img=(900, 1600), rays=(450, 800),
                 n_samples=128, focal=1266.0, cam_z=1.5),
    # cfg4: SemanticKITTI-like mono 370x1220, eval lattice 176x608, 257x257x33, 256 samples
    "cfg4": dict(mapping=dict(nonlinear_mode='linear', h_size=[256, 0], h_range=[51.2, 0], h_half=True,
                              w_size=[128, 0], w_range=[25.6, 0], w_half=False, d_size=[32, 0],
                              d_range=[-2.0, 4.4, 4.4]),
                 aabb=(-25.6, 0.0, -2.0, 25.6, 51.2, 4.4), n_cams=1, img=(370, 1220), rays=(176, 608),
                 n_samples=256, focal=707.0, cam_z=1.7, cam_xy=(0.0, 0.5), yaw0=90.0),
    # cfg5: shipped nuscenes_occ shapes: 257x257x25, centred box, train lattice 48x100, 256 samples
    "cfg5": dict(mapping=dict(nonlinear_mode='linear', h_size=[128, 0], h_range=[40.0, 0], h_half=False,
                              w_size=[128, 0], w_range=[40.0, 0], w_half=False, d_size=[24, 0],
                              d_range=[-1.0, 5.4, 5.4]),
                 aabb=(-40.0, -40.0, -1.0, 40.0, 40.0, 5.4), n_cams=6, img=(768, 1600), rays=(48, 100),
                 n_samples=256, focal=1266.0, cam_z=1.5, cam_xy=(0.0, 0.0)),
}


def make_mapping(name):
    return GridMeterMapping(**CONFIGS[name]["mapping"])


def grid_points_meter(mapping):
    H, W, D = mapping.size_h, mapping.size_w, mapping.size_d
    g = torch.stack(torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                                   torch.arange(D, dtype=torch.float32), indexing='ij'), dim=-1)
    return mapping.grid2meter(g)  # (H, W, D, 3) metres (x, y, z)


def make_volume(name, n_rgb=0, n_sem=0, feat_dtype=torch.float32, seed=0, n_boxes=40, noise=0.05):
    """SDF = distance to (random boxes U ground plane) + N(0, noise^2); logits ~ N(0,1)."""
    cfg = CONFIGS[name]
    mapping = make_mapping(name)
    gen = torch.Generator().manual_seed(seed)
    xyz = grid_points_meter(mapping)
    lo = torch.tensor(cfg["aabb"][:3]); hi = torch.tensor(cfg["aabb"][3:])
    ext = hi - lo
    ground = lo[2] + 0.15 * ext[2]
    sdf = xyz[..., 2] - ground
    cam_xy = torch.tensor(cfg.get("cam_xy_abs", [float(lo[0] + 0.5 * ext[0]), float(lo[1] + 0.5 * ext[1])]))
    if "cam_xy" in cfg:
        cam_xy = torch.tensor(cfg["cam_xy"])
    for _ in range(n_boxes):
        c = lo + torch.rand(3, generator=gen) * ext
        half = torch.tensor([0.5, 0.5, 0.3]) + torch.rand(3, generator=gen) * torch.tensor([3.0, 3.0, 2.0]) * (ext[0] / 80.0)
        c[2] = ground + half[2]
        if torch.linalg.norm(c[:2] - cam_xy) < 0.06 * float(ext[0]) + float(half[:2].max()):
            continue  # keep the camera rig in free space
        q = (xyz - c).abs() - half
        box = torch.linalg.norm(q.clamp_min(0.0), dim=-1) + q.max(dim=-1).values.clamp_max(0.0)
        sdf = torch.minimum(sdf, box)
    sdf = sdf + noise * torch.randn(sdf.shape, generator=gen)
    feat = None
    if n_rgb + n_sem > 0:
        F = SDFVolume.feat_width(n_rgb, n_sem)
        feat = torch.zeros(*sdf.shape, F)
        feat[..., :n_rgb + n_sem] = torch.randn(*sdf.shape, n_rgb + n_sem, generator=gen)
        feat = feat.to(feat_dtype)
    return SDFVolume(mapping, sdf.contiguous().float(), feat, n_rgb, n_sem)


def make_cameras(name, seed=0):
    """(n_cams, 4, 4) float32 img2lidar: (u*t, v*t, t, 1) -> world, pinholes at yaw steps."""
    cfg = CONFIGS[name]
    rng = np.random.RandomState(seed)
    n = cfg["n_cams"]
    Himg, Wimg = cfg["img"]
    f = cfg["focal"]
    K = np.array([[f, 0, Wimg / 2.0, 0], [0, f, Himg / 2.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    lo = np.array(cfg["aabb"][:3]); hi = np.array(cfg["aabb"][3:])
    cxy = np.array(cfg["cam_xy"]) if "cam_xy" in cfg else (lo[:2] + hi[:2]) / 2
    mats = []
    for i in range(n):
        yaw = math.radians(cfg.get("yaw0", 0.0) + 360.0 / n * i)
        # camera axes in world: z forward (cos yaw, sin yaw, 0), x right, y down
        fwd = np.array([math.cos(yaw), math.sin(yaw), 0.0])
        right = np.array([math.sin(yaw), -math.cos(yaw), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        R = np.stack([right, down, fwd], axis=1)
        t = np.array([cxy[0], cxy[1], cfg["cam_z"]]) + (rng.uniform(-1, 1, 3) * np.array([1.0, 1.0, 0.1]) if n > 1 else 0)
        c2w = np.eye(4); c2w[:3, :3] = R; c2w[:3, 3] = t
        mats.append(c2w @ np.linalg.inv(K))
    return torch.tensor(np.stack(mats), dtype=torch.float32)


def make_rays(name, seed=0):
    cfg = CONFIGS[name]
    ny, nx = cfg["rays"]
    Himg, Wimg = cfg["img"]
    return RaySet(img2lidar=make_cameras(name, seed), nx=nx, ny=ny, sx=Wimg / nx, sy=Himg / ny)


def explicit_rays(rays: RaySet):
    """Expand a pixel-lattice RaySet exactly the way the reference does it with torch ops
    (ray_sampler.py:23-31, img2lidar.py:58-69, neus_head.py:321-327)."""
    M = rays.img2lidar
    xs = torch.arange(rays.nx, dtype=torch.float) * rays.sx + rays.ox
    ys = torch.arange(rays.ny, dtype=torch.float) * rays.sy + rays.oy
    pix = torch.stack([xs[None].expand(rays.ny, -1), ys[:, None].expand(-1, rays.nx)], -1).flatten(0, 1)
    pad = torch.cat([pix, torch.ones_like(pix[..., :1])], -1)
    direction = torch.matmul(M[:, None, :3, :3], pad[None, :, :, None]).squeeze(-1)  # n_cams, R, 3
    origin = M[:, None, :3, 3].expand(-1, pix.shape[0], -1)
    direction = direction.flatten(0, 1)
    norm = torch.norm(direction, dim=-1, keepdim=True)
    return RaySet(origins=origin.flatten(0, 1).contiguous(), dirs=(direction / norm).contiguous(),
                  dir_norm=norm.squeeze(-1).contiguous())


def make_render_config(name, **kw):
    cfg = CONFIGS[name]
    return RenderConfig(aabb=cfg["aabb"], n_samples=cfg["n_samples"], **kw)
