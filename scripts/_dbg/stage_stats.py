import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from selfocc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "scripts", "_dbg", "lib_stats.so")
from selfocc_amd import synthetic as sy
from selfocc_amd.render import render_rays, RaySet
d = torch.device("cuda:0")
rays = sy.make_rays("cfg2")
rg = RaySet(img2lidar=rays.img2lidar.to(d), nx=rays.nx, ny=rays.ny, sx=rays.sx, sy=rays.sy)
vol = sy.make_volume("cfg2", n_rgb=3, n_sem=21).to(d)
out = render_rays(vol, rg, sy.make_render_config("cfg2"))
torch.cuda.synchronize()
st = (C.c_ulonglong * 2)()
print("rc", _lib.lib().selfocc_debug_stage_stats(st), "staged", st[0], "fallback", st[1], "frac staged", st[0] / max(1, st[0] + st[1]))
print("wave-steps / (waves*128):", (st[0] + st[1]) / (6 * 29 * 50 * 4 * 128))
