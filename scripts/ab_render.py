"""dev: A/B timing of the cfg2 render launch under the fast-path switches (one process, interleaved rounds).
usage: python scripts/ab_render.py [channels ...]     env SELFOCC_HIP_LIB selects the library build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays, RaySet

d = torch.device("cuda:0")
chans = [int(c) for c in sys.argv[1:]] or [1, 4, 25]
rays = sy.make_rays("cfg2")
rg = RaySet(img2lidar=rays.img2lidar.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy)
VARIANTS = {"default": {}, "no_skip": dict(skip=False), "no_face_safe": dict(face_safe=False),
            "no_skip_no_face_safe": dict(skip=False, face_safe=False), "inv_s_200": dict(inv_s=200.0),
            "no_ahead": dict(ahead=False), "no_ahead_no_face_safe": dict(ahead=False, face_safe=False),
            "no_ahead_inv_s_200": dict(ahead=False, inv_s=200.0)}
for c in chans:
    nr, ns = {1: (0, 0), 4: (3, 0), 25: (3, 21)}[c]
    vol = sy.make_volume("cfg2", n_rgb=nr, n_sem=ns).to(d)
    cfgs = {k: sy.make_render_config("cfg2", **{**dict(inv_s=20.0), **kw}) for k, kw in VARIANTS.items()}
    if c != 1:
        cfgs = {k: v for k, v in cfgs.items() if k in ("default", "no_face_safe")}
    outs = {k: render_rays(vol, rg, cf) for k, cf in cfgs.items()}
    torch.cuda.synchronize()
    times = {k: [] for k in cfgs}
    for rnd in range(5):
        for k, cf in cfgs.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                render_rays(vol, rg, cf, outputs=outs[k])
            b.record(); torch.cuda.synchronize()
            times[k].append(a.elapsed_time(b) / 4)
    for k, t in times.items():
        t = sorted(t)
        print(f"C={c:2d} {k:18s} median {t[len(t)//2]:.4f} ms  min {t[0]:.4f} ms   lib={os.path.basename(os.environ.get('SELFOCC_HIP_LIB', 'default'))}", flush=True)
