"""dev: rays/s of the torch-op CPU port of the reference render (bench.py's cpu_baseline leg) vs torch thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from selfocc_amd import synthetic as sy
from oracle import torch_port as tp
vol = sy.make_volume("cfg2", seed=0); rays = sy.make_rays("cfg2", seed=0); cfg = sy.make_render_config("cfg2", inv_s=20.0)
ex = sy.explicit_rays(rays); dc = vol.to_reference_layout()
print("cpu count", os.cpu_count(), "default torch threads", torch.get_num_threads())
for th in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(th)
    tp.render_port(vol.mapping, dc, 0, 0, ex.origins[:2000], ex.dirs[:2000], ex.dir_norm[:2000], cfg)
    c0 = time.perf_counter()
    tp.render_port(vol.mapping, dc, 0, 0, ex.origins[:90000], ex.dirs[:90000], ex.dir_norm[:90000], cfg, chunk=90000)
    dt = time.perf_counter() - c0
    print(f"threads {th:4d}: {90000 / dt:10.0f} rays/s  ({dt:.2f} s per 90 000-ray chunk)", flush=True)
