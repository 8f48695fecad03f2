import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, oracle
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays, RaySet
vol = sy.make_volume("cfg1", seed=9)
rays = sy.make_rays("cfg1", seed=9)
M = rays.img2lidar.clone().repeat(3, 1, 1)
M[0, :3, 3] += torch.tensor([-9.0, 0.3, 0.2]); M[1, :3, 3] += torch.tensor([-3.0, -8.5, 0.4]); M[2, :3, 3] += torch.tensor([-30.0, 40.0, 9.0])
rays.img2lidar = M
cfg = sy.make_render_config("cfg1", inv_s=20.0)
ref = oracle.render_fwd(vol, rays, cfg, per_sample=True)
d = torch.device("cuda:0")
got = render_rays(vol.to(d), RaySet(img2lidar=M.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy), cfg, per_sample=True)
for k in ('deltas', 'ts', 'nears', 'fars'):
    g, r = got[k].cpu(), ref[k]
    diff = (g - r).abs()
    i = diff.flatten().argmax()
    print(k, 'max diff', diff.max().item(), 'at', i.item(), 'got', g.flatten()[i].item(), 'ref', r.flatten()[i].item(), 'nan', torch.isnan(g).sum().item(), torch.isnan(r).sum().item())
j = (got['deltas'].cpu() - ref['deltas']).abs().amax(1).argmax()
print('ray', j.item(), 'near/far ref', ref['nears'][j].item(), ref['fars'][j].item(), 'got', got['nears'][j].item(), got['fars'][j].item())
