"""dev: launch the cfg2 render kernel a few times (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays, RaySet
c = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = torch.device("cuda:0")
rays = sy.make_rays("cfg2")
rg = RaySet(img2lidar=rays.img2lidar.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy)
nr, ns = {1: (0, 0), 4: (3, 0), 25: (3, 21)}[c]
vol = sy.make_volume("cfg2", n_rgb=nr, n_sem=ns).to(d)
cfg = sy.make_render_config("cfg2")
out = render_rays(vol, rg, cfg)
for _ in range(n):
    render_rays(vol, rg, cfg, outputs=out)
torch.cuda.synchronize()
