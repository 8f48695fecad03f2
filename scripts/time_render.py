"""Quick on-GPU timing of selfocc_render_fwd on BASELINE cfg2 (dev tool, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays, RaySet

d = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
rays = sy.make_rays(name)
rg = RaySet(img2lidar=rays.img2lidar.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy)
ex = sy.explicit_rays(rays)
re = RaySet(origins=ex.origins.to(d), dirs=ex.dirs.to(d), dir_norm=ex.dir_norm.to(d))
cfg = sy.make_render_config(name)
for (n_rgb, n_sem, dt) in [(0, 0, torch.float32), (3, 0, torch.float32), (3, 0, torch.bfloat16),
                           (3, 21, torch.float32), (3, 21, torch.bfloat16)]:
    vol = sy.make_volume(name, n_rgb=n_rgb, n_sem=n_sem, feat_dtype=dt).to(d)
    for label, r in [("pixgrid", rg), ("explicit", re)]:
        out = render_rays(vol, r, cfg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            render_rays(vol, r, cfg, outputs=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name} C={1+n_rgb+n_sem:2d} {str(dt):15s} {label:9s} {ms:8.3f} ms  {r.n_rays/ms/1e3:9.1f} Mrays/s  acc_mean={out['acc'].mean().item():.3f}", flush=True)
