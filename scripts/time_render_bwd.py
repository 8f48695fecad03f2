"""dev: HIP-event time of selfocc_render_bwd alone at the shipped nuscenes_occ training launch (257x257x25,
25-channel volume, 6 x 48x100 rays x 256 samples), per scatter mode.  argv: modes (default: atomic binned)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays_autograd, RaySet, SDFVolume
d = torch.device("cuda:0")
n_sem = int(os.environ.get("SO_NSEM", "21"))
n_rgb = 3 if n_sem >= 0 else 0
n_sem = max(n_sem, 0)
rays = sy.make_rays("cfg5")
rg = RaySet(img2lidar=rays.img2lidar.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy)
vol = sy.make_volume("cfg5", n_rgb=n_rgb, n_sem=n_sem).to(d)
res = {}
for mode in (sys.argv[1:] or ["atomic", "binned"]):
    cfg = sy.make_render_config("cfg5")
    cfg.bwd_scatter = mode
    inv_s = torch.tensor([float(cfg.inv_s)], device=d, requires_grad=True)
    ts = []
    for it in range(8):
        sdf = vol.sdf.detach().clone().requires_grad_(True)
        feat = None if vol.feat is None else vol.feat.detach().clone().requires_grad_(True)
        out = render_rays_autograd(SDFVolume(vol.mapping, sdf, feat, vol.n_rgb, vol.n_sem), inv_s, rg, cfg)
        loss = out['depth'].mean() + out['sdf'].abs().mean() * 0.1 + (out['grad'].norm(dim=-1) - 1).square().mean() * 0.1
        if n_rgb:
            loss = loss + out['rgb'].mean() + out['sem'].square().mean()
        g = torch.autograd.grad(loss, [out[k] for k in ('depth', 'sdf', 'grad') + (('rgb', 'sem') if n_rgb else ())], retain_graph=True)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.autograd.backward([out[k] for k in ('depth', 'sdf', 'grad') + (('rgb', 'sem') if n_rgb else ())], g)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res[mode] = round(sorted(ts[2:])[len(ts[2:]) // 2], 3)
print(json.dumps(dict(render_bwd_ms=res, n_sem=n_sem)))
