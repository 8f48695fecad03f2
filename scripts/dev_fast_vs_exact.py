"""dev: deviation statistics of the FAST render path vs EXACT on the GPU (cfg2, C=25)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays, RaySet
from util import cell_margin
d = torch.device("cuda:0")
rays = sy.make_rays("cfg2")
rg = RaySet(img2lidar=rays.img2lidar.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy)
ex = sy.explicit_rays(rays)
exd = RaySet(origins=ex.origins.to(d), dirs=ex.dirs.to(d), dir_norm=ex.dir_norm.to(d))
vol = sy.make_volume("cfg2", n_rgb=3, n_sem=21).to(d)
for inv_s in (20.0, 200.0):
    cfg = sy.make_render_config("cfg2", inv_s=inv_s)
    fast = render_rays(vol, rg, cfg)
    exact = render_rays(vol, rg, sy.make_render_config("cfg2", inv_s=inv_s, exact=True))
    torch.cuda.synchronize()
    margin = cell_margin(vol.mapping, exd, cfg, exact['nears'], exact['fars'])
    for thr in (1e-4, 1e-3):
        ok = (exact['acc'] > 0.05) & (margin > thr)
        print(f"inv_s={inv_s} margin>{thr}: kept={ok.float().mean().item():.4f} (acc>0.05: {(exact['acc']>0.05).float().mean().item():.4f})")
        for k in ('depth', 'acc', 'rgb', 'sem', 'max_depth'):
            f, e = fast[k], exact[k]
            rel = (f - e).abs() / (e.abs() + 1e-5)
            m = ok if rel.dim() == 1 else ok[:, None].expand_as(rel)
            print(f"  {k:10s} max rel {rel[m].max().item():.3e}  frac>1e-4 {(rel[m] > 1e-4).float().mean().item():.3e}  max abs {(f-e).abs()[m].max().item():.3e}")
