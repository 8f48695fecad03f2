"""dev: binned vs atomic render backward on small and training-size launches: relative differences per gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd import abi, synthetic as sy
from selfocc_amd.render import render_rays_autograd, RaySet, SDFVolume
d = torch.device("cuda:0")
def run(name, n_rgb, n_sem, S, explicit):
    rays = sy.make_rays(name, seed=11)
    vol = sy.make_volume(name, n_rgb=n_rgb, n_sem=n_sem, seed=11).to(d)
    if explicit:
        ex = sy.explicit_rays(rays)
        rg = RaySet(origins=ex.origins.to(d), dirs=ex.dirs.to(d), dir_norm=ex.dir_norm.to(d))
    else:
        rg = RaySet(img2lidar=rays.img2lidar.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy)
    res = {}
    for mode in ("atomic", "binned"):
        cfg = sy.make_render_config(name)
        cfg.n_samples = S
        cfg.bwd_scatter = mode
        inv_s = torch.tensor([float(cfg.inv_s)], device=d, requires_grad=True)
        sdf = vol.sdf.detach().clone().requires_grad_(True)
        feat = None if vol.feat is None else vol.feat.detach().clone().requires_grad_(True)
        out = render_rays_autograd(SDFVolume(vol.mapping, sdf, feat, n_rgb, n_sem), inv_s, rg, cfg)
        loss = out['depth'].mean() + out['sdf'].abs().mean() * 0.1 + (out['grad'].norm(dim=-1) - 1).square().mean() * 0.1
        if n_rgb: loss = loss + out['rgb'].mean()
        if n_sem: loss = loss + out['sem'].square().mean()
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (sdf.grad, None if feat is None else feat.grad, inv_s.grad)
    a, b = res["atomic"], res["binned"]
    rel = lambda x, y: ((x - y).norm() / (y.norm() + 1e-30)).item()
    print(name, n_rgb, n_sem, S, "explicit" if explicit else "grid", "sdf rel", rel(b[0], a[0]), "sum", a[0].sum().item(), b[0].sum().item(),
          "feat rel", None if a[1] is None else rel(b[1], a[1]), "inv_s", a[2].item(), b[2].item(), flush=True)
for args in (("cfg1", 0, 0, 32, True), ("cfg1", 3, 0, 32, True), ("cfg1", 3, 5, 32, False), ("cfg1", 3, 21, 100, True), ("cfg1", 3, 17, 300, True),
             ("cfg5", 0, 0, 256, False), ("cfg5", 3, 21, 256, False)):
    run(*args)
