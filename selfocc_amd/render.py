"""Host side of the fused SDF ray-march renderer (selfocc_render_fwd / _bwd).

This is the operator-level surface the head (selfocc_amd/model/head) builds on; it
replaces what the reference obtains from the sdfstudio fork's
``NeuSCustomModel.__call__(RayBundle)`` (model/head/neus_head/neus_head.py:353,394,531).
Tensors are only device-memory handles here: the arithmetic is in csrc/render_*.hip.
"""
import os
from dataclasses import dataclass, field
from typing import Optional

import torch

from . import abi
from ._lib import lib, check, ptr, current_stream


@dataclass
class SDFVolume:
    """The pre-computed field volume in the layout the kernels read.

    sdf  : (H, W, D) float32
    feat : (H, W, D, F) float32 / bfloat16 or None; channels = raw colour (3) then
           semantic logits (n_sem); F == 4 when n_sem == 0 (one pad channel).
    The reference materialises (1, 1 + color_dims, H, W, D) instead
    (nerfacc_head/bev_nerf.py:74-95); ``from_reference_layout`` converts.
    """
    mapping: object            # GridMeterMapping / LinearMapping (has to_abi())
    sdf: torch.Tensor
    feat: Optional[torch.Tensor] = None
    n_rgb: int = 0
    n_sem: int = 0

    @staticmethod
    def feat_width(n_rgb, n_sem):
        if n_rgb + n_sem == 0:
            return 0
        return 4 if n_sem == 0 else n_rgb + n_sem

    @classmethod
    def from_reference_layout(cls, mapping, density_color, n_rgb=0, n_sem=0, feat_dtype=torch.float32):
        """density_color: (1, 1 + n_rgb + n_sem, H, W, D) as in the reference."""
        assert density_color.dim() == 5 and density_color.shape[0] == 1
        assert density_color.shape[1] == 1 + n_rgb + n_sem
        sdf = density_color[0, 0].contiguous().float()
        feat = None
        if n_rgb + n_sem > 0:
            F = cls.feat_width(n_rgb, n_sem)
            H, W, D = sdf.shape
            feat = torch.zeros(H, W, D, F, dtype=feat_dtype, device=sdf.device)
            feat[..., :n_rgb + n_sem] = density_color[0, 1:].permute(1, 2, 3, 0).to(feat_dtype)
        return cls(mapping, sdf, feat, n_rgb, n_sem)

    def to_reference_layout(self):
        ch = [self.sdf[None]]
        if self.feat is not None:
            ch.append(self.feat[..., :self.n_rgb + self.n_sem].float().permute(3, 0, 1, 2))
        return torch.cat(ch, 0)[None]

    def cpu(self):
        return SDFVolume(self.mapping, self.sdf.cpu(), None if self.feat is None else self.feat.cpu(),
                         self.n_rgb, self.n_sem)

    def to(self, device):
        return SDFVolume(self.mapping, self.sdf.to(device), None if self.feat is None else self.feat.to(device),
                         self.n_rgb, self.n_sem)


@dataclass
class RaySet:
    """Either explicit rays (RayBundle-like) or a per-camera pixel lattice from which
    the kernel generates origin / direction itself (RaySampler + Img2LiDAR fused)."""
    origins: Optional[torch.Tensor] = None    # (N, 3)
    dirs: Optional[torch.Tensor] = None       # (N, 3) unit
    dir_norm: Optional[torch.Tensor] = None   # (N,)
    img2lidar: Optional[torch.Tensor] = None  # (n_cams, 4, 4) float32
    nx: int = 0
    ny: int = 0
    sx: float = 1.0
    sy: float = 1.0
    ox: float = 0.0
    oy: float = 0.0

    @property
    def pixel_grid(self):
        return self.img2lidar is not None

    @property
    def n_rays(self):
        if self.pixel_grid:
            return self.img2lidar.shape[0] * self.nx * self.ny
        return self.origins.shape[0]

    @property
    def device(self):
        return (self.img2lidar if self.pixel_grid else self.origins).device

    def cpu(self):
        c = lambda t: None if t is None else t.cpu()
        return RaySet(c(self.origins), c(self.dirs), c(self.dir_norm), c(self.img2lidar),
                      self.nx, self.ny, self.sx, self.sy, self.ox, self.oy)


@dataclass
class RenderConfig:
    aabb: tuple                       # (xmin, ymin, zmin, xmax, ymax, zmax)
    n_samples: int = 128
    inv_s: float = 20.0
    inv_s_dev: object = None          # optional 1-element float32 DEVICE tensor: the kernels read inv_s from it (no host copy)
    near_plane: float = 0.0
    sample_pos: int = abi.SAMPLE_AT_START
    jitter_mode: int = abi.JITTER_NONE
    bkgd_mode: int = abi.BKGD_NONE
    bkgd: tuple = (0.0, 0.0, 0.0)
    depth_div_norm: bool = True
    clamp_rgb: bool = False
    exact: bool = False               # canonical IEEE op order (bit-exact with the oracle), slower
    brick: bool = True                # fast path: re-pack the SDF volume into 8-corner records per launch
    skip: bool = True                 # fast path + brick: composite saturated free-space samples without interpolating
    face_safe: bool = True            # fast path: canonical cell selection within a few ulp of a voxel face (~6 % slower)
    ahead: bool = True                # SDF-only per-ray launches with brick + skip: code-ahead skip marcher (A/B)
    ray_per_lane: bool = False        # ignored since ABI 30 (was: per-sample launches through ray-per-lane kernels, an A/B switch)
    bwd_scatter: str = 'auto'         # backward: 'binned' = brick-binned LDS scatter of the volume gradients (needs a scratch
                                      # of ~(record + 8) bytes per sample), 'atomic' = per-sample row atomics, 'auto' = binned
                                      # from 2^19 samples per launch — measured cross-over 0.4 - 0.5 M samples at 25 channels, scripts/time_render_bwd_sizes.py
                                      # (SELFOCC_RB_SCATTER overrides 'auto')


def _c(t, dtype=torch.float32):
    assert t.dtype == dtype and t.is_contiguous(), f"need contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}"
    return t


def marshal_render_args(vol: SDFVolume, rays: RaySet, cfg: RenderConfig, *, per_sample=False,
                        want_grad_samples=False, t_rand=None, bkgd_rays=None, outputs=None):
    """Fill an ``so_render_args`` for tensors living on ONE device (CUDA for the HIP
    library; the test oracle feeds CPU tensors through the same marshalling).
    Returns (args, outputs_dict, keepalive)."""
    dev = vol.sdf.device
    a = abi.SoRenderArgs()
    a.map = vol.mapping.to_abi()
    H, W, D = a.map.h.tot_len, a.map.w.tot_len, a.map.d.tot_len
    assert tuple(vol.sdf.shape) == (H, W, D), f"sdf volume {tuple(vol.sdf.shape)} != mapping {(H, W, D)}"
    a.sdf_vol = ptr(_c(vol.sdf))
    a.n_rgb, a.n_sem = vol.n_rgb, vol.n_sem
    if vol.feat is not None:
        assert vol.feat.is_contiguous() and vol.feat.shape[:3] == (H, W, D)
        assert vol.feat.dtype in (torch.float32, torch.bfloat16)
        a.feat_vol = ptr(vol.feat)
        a.feat_dtype = abi.DTYPE_F32 if vol.feat.dtype == torch.float32 else abi.DTYPE_BF16
        a.feat_stride = vol.feat.shape[3]
    N = rays.n_rays
    a.n_rays = N
    keep = [vol, rays, t_rand, bkgd_rays]
    if rays.pixel_grid:
        a.ray_mode = abi.RAYS_PIXEL_GRID
        a.img2lidar = ptr(_c(rays.img2lidar))
        a.n_cams, a.nx, a.ny = rays.img2lidar.shape[0], rays.nx, rays.ny
        a.sx, a.sy, a.ox, a.oy = rays.sx, rays.sy, rays.ox, rays.oy
    else:
        a.ray_mode = abi.RAYS_EXPLICIT
        a.origins, a.dirs = ptr(_c(rays.origins)), ptr(_c(rays.dirs))
        a.dir_norm = ptr(None if rays.dir_norm is None else _c(rays.dir_norm))
    for i in range(6):
        a.aabb[i] = float(cfg.aabb[i])
    a.near_plane = cfg.near_plane
    a.n_samples = S = cfg.n_samples
    a.sample_pos = cfg.sample_pos
    a.jitter_mode = cfg.jitter_mode
    if cfg.jitter_mode != abi.JITTER_NONE:
        exp = (N,) if cfg.jitter_mode == abi.JITTER_SINGLE else (N, S + 1)
        assert t_rand is not None and tuple(t_rand.shape) == exp, f"t_rand must be {exp}"
        a.t_rand = ptr(_c(t_rand))
    a.inv_s = float(cfg.inv_s)
    if cfg.inv_s_dev is not None:
        t = cfg.inv_s_dev
        assert t.dtype == torch.float32 and t.numel() == 1 and t.device == dev, "inv_s_dev: one float32 on the volume's device"
        a.inv_s_dev = ptr(t)
        keep.append(t)
    a.bkgd_mode = cfg.bkgd_mode
    for i in range(3):
        a.bkgd[i] = float(cfg.bkgd[i])
    if cfg.bkgd_mode == abi.BKGD_PER_RAY:
        assert bkgd_rays is not None and tuple(bkgd_rays.shape) == (N, 3)
        a.bkgd_rays = ptr(_c(bkgd_rays))
    a.flags = (abi.FLAG_DEPTH_DIV_NORM if cfg.depth_div_norm else 0) | (abi.FLAG_CLAMP_RGB if cfg.clamp_rgb else 0) | \
        (abi.FLAG_EXACT if cfg.exact else 0) | (0 if cfg.skip else abi.FLAG_NO_SKIP) | \
        (0 if cfg.face_safe else abi.FLAG_NO_FACE_SAFE) | (0 if cfg.ahead else abi.FLAG_NO_AHEAD) | \
        (abi.FLAG_RAY_PER_LANE if cfg.ray_per_lane else 0)

    f32 = dict(dtype=torch.float32, device=dev)
    out = outputs if outputs is not None else {}
    def alloc(name, *shape):
        if name not in out:
            out[name] = torch.empty(*shape, **f32)
        return ptr(out[name])
    a.depth, a.acc = alloc('depth', N), alloc('acc', N)
    a.max_depth, a.nears, a.fars = alloc('max_depth', N), alloc('nears', N), alloc('fars', N)
    if vol.n_rgb == 3:
        a.rgb = alloc('rgb', N, 3)
    if vol.n_sem > 0:
        a.sem = alloc('sem', N, vol.n_sem)
    if per_sample:
        a.weights, a.ts, a.deltas = alloc('weights', N, S), alloc('ts', N, S), alloc('deltas', N, S)
        if want_grad_samples:
            a.sdf, a.grad = alloc('sdf', N, S), alloc('grad', N, S, 3)
    keep.append(out)
    return a, out, keep


def render_rays(vol: SDFVolume, rays: RaySet, cfg: RenderConfig, *, per_sample=False,
                want_grad_samples=False, t_rand=None, bkgd_rays=None, outputs=None):
    """Forward render on the GPU (no autograd).  Returns a dict of per-ray tensors
    (depth, acc, rgb, sem, max_depth, nears, fars) and, with ``per_sample``, the
    (N, S) tensors weights / ts / deltas (/ sdf / grad)."""
    if not vol.sdf.is_cuda:
        raise RuntimeError("selfocc_amd.render_rays needs CUDA(HIP) tensors: there is no CPU fallback")
    a, out, _keep = marshal_render_args(vol, rays, cfg, per_sample=per_sample,
                                        want_grad_samples=want_grad_samples, t_rand=t_rand,
                                        bkgd_rays=bkgd_rays, outputs=outputs)
    if cfg.brick and not cfg.exact and rays.n_rays * cfg.n_samples >= 16 * vol.sdf.numel():
        a.sdf_brick = ptr(_brick_workspace(vol.sdf))   # the re-pack only pays off for large ray batches
    check(lib().selfocc_render_fwd(a, current_stream(vol.sdf.device)), "selfocc_render_fwd")
    return out


_BRICK_WS = {}


def _brick_workspace(sdf):
    """Per (device, stream, shape) scratch for the 8-corner records of the SDF volume ([H][W][D][8] f32) followed
    by one skip-code byte per cell; rewritten by every launch on the launch stream, so it is never stale and two
    streams never share one."""
    key = (sdf.device, torch.cuda.current_stream(sdf.device).cuda_stream, tuple(sdf.shape))
    ws = _BRICK_WS.get(key)
    if ws is None:
        n = sdf.numel()
        ws = _BRICK_WS[key] = torch.empty((n * 33 + 15) // 16 * 16, dtype=torch.uint8, device=sdf.device)
    return ws


def _scatter_workspace(device, nbytes):
    """Scratch of the binned backward scatter (~(record + 8) bytes per sample: 0.95 GB at the nuscenes_occ training shape,
    28 800 rays x 256 samples x 128-byte records).  Taken from torch's caching allocator for the duration of ONE backward
    call — stream-ordered, returned to the pool when the call ends, so that the forward pass of the next iteration can
    reuse the memory (until round 4 it was a grow-only, process-lifetime buffer per (device, stream): +0.95 GB of peak)."""
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


class _RenderFunction(torch.autograd.Function):
    """Differentiable wrt the SDF volume, the feature volume and inv_s (a 0-dim / 1-elem
    tensor).  Ray geometry carries no gradient — the reference's rays come from constant
    camera matrices (model/head/nerfacc_head/img2lidar.py:36-69)."""

    @staticmethod
    def forward(ctx, sdf_vol, feat_vol, inv_s, mapping, n_rgb, n_sem, rays, cfg, t_rand, bkgd_rays, want_grad_samples):
        vol = SDFVolume(mapping, sdf_vol, feat_vol, n_rgb, n_sem)
        # the kernels read inv_s from the device tensor itself: no read-back, no stream sync, nothing to go stale
        cfg_run = RenderConfig(**{**cfg.__dict__, 'inv_s_dev': inv_s.detach().reshape(1).float().contiguous()})
        out = render_rays(vol, rays, cfg_run, per_sample=True, want_grad_samples=want_grad_samples,
                          t_rand=t_rand, bkgd_rays=bkgd_rays)
        # tensor inputs go through save_for_backward (version-counter checks, saved-tensor hooks); the ray set and
        # the config are plain python state
        ctx.has = (feat_vol is not None, t_rand is not None, bkgd_rays is not None)
        ctx.save_for_backward(*[t for t in (sdf_vol, feat_vol, t_rand, bkgd_rays, cfg_run.inv_s_dev) if t is not None])
        ctx.meta = (mapping, n_rgb, n_sem)
        ctx.rays, ctx.cfg = rays, RenderConfig(**{**cfg_run.__dict__, 'inv_s_dev': None})
        ctx.inv_s_shape = inv_s.shape
        ctx.keys = ['depth', 'acc'] + (['rgb'] if n_rgb else []) + (['sem'] if n_sem else []) + ['weights'] + \
            (['sdf', 'grad'] if want_grad_samples else [])
        ctx.nondiff_keys = ['ts', 'deltas', 'max_depth', 'nears', 'fars']
        nd = tuple(out[k] for k in ctx.nondiff_keys)
        ctx.mark_non_differentiable(*nd)
        return tuple(out[k] for k in ctx.keys) + nd

    @staticmethod
    def backward(ctx, *grads):
        saved = list(ctx.saved_tensors)
        sdf_vol = saved.pop(0)
        feat_vol = saved.pop(0) if ctx.has[0] else None
        t_rand = saved.pop(0) if ctx.has[1] else None
        bkgd_rays = saved.pop(0) if ctx.has[2] else None
        inv_s_dev = saved.pop(0)
        mapping, n_rgb, n_sem = ctx.meta
        vol = SDFVolume(mapping, sdf_vol, feat_vol, n_rgb, n_sem)
        rays, cfg = ctx.rays, RenderConfig(**{**ctx.cfg.__dict__, 'inv_s_dev': inv_s_dev})
        gmap = {k: g for k, g in zip(ctx.keys, grads)}
        a, _out, _keep = marshal_render_args(vol, rays, cfg, per_sample=False, t_rand=t_rand,
                                             bkgd_rays=bkgd_rays, outputs={})
        ba = abi.SoRenderBwdArgs()
        ba.fwd = a
        hold = []
        def gp(name):
            g = gmap.get(name)
            if g is None:
                return None
            g = g.contiguous().float()
            hold.append(g)
            return ptr(g)
        ba.g_depth, ba.g_acc, ba.g_rgb, ba.g_sem = gp('depth'), gp('acc'), gp('rgb'), gp('sem')
        ba.g_weights, ba.g_sdf, ba.g_grad = gp('weights'), gp('sdf'), gp('grad')
        g_sdf_vol = torch.zeros_like(vol.sdf)
        ba.g_sdf_vol = ptr(g_sdf_vol)
        g_feat = None
        if vol.feat is not None and ctx.needs_input_grad[1]:
            g_feat = torch.zeros(vol.feat.shape, dtype=torch.float32, device=vol.feat.device)
            ba.g_feat_vol = ptr(g_feat)
        g_inv_s = torch.zeros(1, device=vol.sdf.device)
        ba.g_inv_s = ptr(g_inv_s)
        mode = cfg.bwd_scatter if cfg.bwd_scatter != 'auto' else os.environ.get('SELFOCC_RB_SCATTER', 'auto')
        if mode == 'binned' or (mode == 'auto' and rays.n_rays * cfg.n_samples >= (1 << 19)):
            need = int(lib().selfocc_render_bwd_ws_bytes(ba))
            if need > 0:
                ws = _scatter_workspace(vol.sdf.device, need)
                ba.scatter_ws, ba.scatter_ws_bytes = ptr(ws), ws.numel()
                hold.append(ws)
            elif mode == 'binned':
                raise RuntimeError("bwd_scatter='binned': the volume / launch is outside the binned scatter's range")
        check(lib().selfocc_render_bwd(ba, current_stream(vol.sdf.device)), "selfocc_render_bwd")
        if g_feat is not None and vol.feat.dtype != torch.float32:
            g_feat = g_feat.to(vol.feat.dtype)
        return (g_sdf_vol, g_feat, g_inv_s.reshape(ctx.inv_s_shape), None, None, None, None, None, None, None, None)


def render_rays_autograd(vol: SDFVolume, inv_s: torch.Tensor, rays: RaySet, cfg: RenderConfig, *,
                         want_grad_samples=True, t_rand=None, bkgd_rays=None):
    """Training-time render: returns the same dict as ``render_rays(per_sample=True)`` with
    depth / acc / rgb / sem / weights / sdf / grad attached to the autograd graph of
    ``vol.sdf``, ``vol.feat`` and ``inv_s``."""
    res = _RenderFunction.apply(vol.sdf, vol.feat, inv_s, vol.mapping, vol.n_rgb, vol.n_sem, rays, cfg,
                                t_rand, bkgd_rays, want_grad_samples)
    keys = ['depth', 'acc'] + (['rgb'] if vol.n_rgb else []) + (['sem'] if vol.n_sem else []) + ['weights'] + \
        (['sdf', 'grad'] if want_grad_samples else []) + ['ts', 'deltas', 'max_depth', 'nears', 'fars']
    return dict(zip(keys, res))
