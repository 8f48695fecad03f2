"""dev: training-shape render forward + backward (cfg5 = shipped nuscenes_occ: 257x257x25, 25-channel volume,
6 x 48x100 rays x 256 samples) a few times, for rocprofv3.  argv[1]: 1 = gradient w.r.t. the feature volume too."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays_autograd, RaySet, SDFVolume
feat_grad = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
d = torch.device("cuda:0")
rays = sy.make_rays("cfg5")
rg = RaySet(img2lidar=rays.img2lidar.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy)
vol = sy.make_volume("cfg5", n_rgb=3, n_sem=21).to(d)
cfg = sy.make_render_config("cfg5")
inv_s = torch.tensor([float(cfg.inv_s)], device=d, requires_grad=True)
for it in range(4):
    sdf = vol.sdf.detach().clone().requires_grad_(True)
    feat = vol.feat.detach().clone().requires_grad_(feat_grad)
    v = SDFVolume(vol.mapping, sdf, feat, vol.n_rgb, vol.n_sem)
    out = render_rays_autograd(v, inv_s, rg, cfg)
    loss = out['depth'].mean() + out['rgb'].mean() + out['sem'].square().mean() + out['sdf'].abs().mean() * 0.1 + \
        (out['grad'].norm(dim=-1) - 1).square().mean() * 0.1
    loss.backward()
torch.cuda.synchronize()
print("ok", float(loss))
