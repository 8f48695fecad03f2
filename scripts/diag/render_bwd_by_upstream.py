"""Diagnostic (GPU): selfocc_render_bwd vs float64 autograd of the port at the nuscenes_occ training shape, ONE upstream
gradient at a time — which of d L / d {depth, acc, weights, sdf, grad, rgb, sem} carries the float32 noise."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_port as tp
from selfocc_amd import abi, synthetic as sy
from selfocc_amd.render import render_rays_autograd, RaySet, SDFVolume

D0 = torch.device("cuda:0")
n_rgb, n_sem = 3, 21
vol = sy.make_volume("cfg5", n_rgb=n_rgb, n_sem=n_sem, seed=3)
full = sy.explicit_rays(sy.make_rays("cfg5", seed=3))
idx = torch.arange(0, full.n_rays, 14)
ex = RaySet(origins=full.origins[idx].contiguous(), dirs=full.dirs[idx].contiguous(), dir_norm=full.dir_norm[idx].contiguous())
cfg = sy.make_render_config("cfg5", inv_s=12.0, jitter_mode=abi.JITTER_SINGLE, bkgd_mode=abi.BKGD_PER_RAY)
N, S = ex.n_rays, cfg.n_samples
g = torch.Generator().manual_seed(4)
t_rand, bk = torch.rand(N, generator=g), torch.rand(N, 3, generator=g)
G = dict(depth=torch.randn(N, generator=g), acc=torch.randn(N, generator=g), weights=torch.randn(N, S, generator=g),
         sdf=0.1 * torch.randn(N, S, generator=g), grad=0.1 * torch.randn(N, S, 3, generator=g),
         rgb=torch.randn(N, 3, generator=g), sem=torch.randn(N, n_sem, generator=g))
dd = torch.float64
vol64 = vol.to_reference_layout()[0].to(dd).requires_grad_(True)
inv_s64 = torch.tensor(cfg.inv_s, dtype=dd, requires_grad=True)
ref = tp.render_port_differentiable(vol.mapping, vol64, n_rgb, n_sem, ex.origins.to(dd), ex.dirs.to(dd), ex.dir_norm.to(dd),
                                    cfg, inv_s64, t_rand.to(dd), bk.to(dd))
if os.environ.get('DIAG_MASK', '1') == '1':      # the test's mask: no upstream gradient on rays with a sample within 1e-4 voxel of a face
    pos = ex.origins.to(dd)[:, None, :] + ex.dirs.to(dd)[:, None, :] * ref['starts'].detach()[..., None]
    gc = vol.mapping.meter2grid(pos)
    fr = gc - torch.floor(gc)
    keep = (torch.minimum(fr, 1 - fr).amin(dim=(1, 2)) > 1e-4)
    for k in G:
        G[k] = G[k] * keep.reshape(-1, *([1] * (G[k].dim() - 1))).to(G[k].dtype)
    G['depth'] = G['depth'] * (ref['acc'].detach() > 0.05).float()
    print('rays with upstream gradient:', keep.float().mean().item())
v = vol.to(D0)
rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
for mode in ("atomic",):
    cfg.bwd_scatter = mode
    sdf_p, feat_p = v.sdf.clone().requires_grad_(True), v.feat.clone().requires_grad_(True)
    inv_s = torch.tensor([cfg.inv_s], device=D0, requires_grad=True)
    out = render_rays_autograd(SDFVolume(v.mapping, sdf_p, feat_p, n_rgb, n_sem), inv_s,
                               RaySet(origins=ex.origins.to(D0), dirs=ex.dirs.to(D0), dir_norm=ex.dir_norm.to(D0)),
                               cfg, want_grad_samples=True, t_rand=t_rand.to(D0), bkgd_rays=bk.to(D0))
    for k in G:
        fwd = ((out[k].detach().cpu().double() - ref[k].detach()).abs().max() / ref[k].detach().abs().max()).item()
        gv, gi = torch.autograd.grad((ref[k] * G[k].to(dd)).sum(), [vol64, inv_s64], retain_graph=True, allow_unused=True)
        hs, hf, hi = torch.autograd.grad((out[k] * G[k].to(D0)).sum(), [sdf_p, feat_p, inv_s], retain_graph=True, allow_unused=True)
        hs = torch.zeros_like(sdf_p) if hs is None else hs
        hf = torch.zeros_like(feat_p) if hf is None else hf
        m = dict(mode=mode, key=k, fwd_max=fwd, g_sdf_norm=gv[0].norm().item(), sdf_l2=rel(hs.cpu().double(), gv[0]),
                 sdf_max=((hs.cpu().double() - gv[0]).abs().max() / gv[0].abs().max().clamp_min(1e-300)).item(),
                 feat_l2=rel(hf.cpu().double(), gv[1:].permute(1, 2, 3, 0)) if gv[1:].norm() > 0 else 0.0,
                 inv_s_ref=0.0 if gi is None else gi.item(), inv_s_got=0.0 if hi is None else hi.item())
        print(json.dumps(m), flush=True)
