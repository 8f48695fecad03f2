"""dev: atomic vs binned render backward at smaller launches (where should 'auto' switch?): 257x257x25 volume, 25 channels,
1 .. 6 cameras x (ny x nx) lattices x 256 samples."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays_autograd, RaySet, SDFVolume
d = torch.device("cuda:0")
rays = sy.make_rays("cfg5")
vol = sy.make_volume("cfg5", n_rgb=3, n_sem=21).to(d)
for cams, ny, nx in ((1, 8, 16), (1, 16, 32), (1, 24, 50), (2, 24, 50), (2, 48, 100), (6, 48, 100)):
    rg = RaySet(img2lidar=rays.img2lidar[:cams].to(d).contiguous(), nx=nx, ny=ny, sx=rays.sx * rays.nx / nx, sy=rays.sy * rays.ny / ny)
    res = {}
    for mode in ("atomic", "binned"):
        cfg = sy.make_render_config("cfg5")
        cfg.bwd_scatter = mode
        inv_s = torch.tensor([float(cfg.inv_s)], device=d, requires_grad=True)
        ts = []
        for it in range(7):
            sdf = vol.sdf.detach().clone().requires_grad_(True)
            feat = vol.feat.detach().clone().requires_grad_(True)
            out = render_rays_autograd(SDFVolume(vol.mapping, sdf, feat, 3, 21), inv_s, rg, cfg)
            keys = ('depth', 'sdf', 'grad', 'rgb', 'sem')
            loss = out['depth'].mean() + out['sdf'].abs().mean() * 0.1 + (out['grad'].norm(dim=-1) - 1).square().mean() * 0.1 + out['rgb'].mean() + out['sem'].square().mean()
            g = torch.autograd.grad(loss, [out[k] for k in keys], retain_graph=True)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.autograd.backward([out[k] for k in keys], g)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = round(sorted(ts[2:])[2], 3)
    print(json.dumps(dict(samples=cams * ny * nx * 256, **res)))
