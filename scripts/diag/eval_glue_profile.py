"""Dev: the torch-side fill / copy / add kernels of one eval frame (nuscenes_depth by default), grouped by parent op chain."""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "scripts"))
import numpy as np, torch
import hotpath_common as hc
from torch.profiler import profile, ProfilerActivity
d = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_depth"
os.environ["eval"] = "true"
torch.manual_seed(0); np.random.seed(0)
cfg = hc.shipped_for_eval(name)
mods = hc.build(cfg, d)
for m in mods[:3]: m.eval()
fr = hc.frame_inputs(cfg, name, d, seed=1, want_images=False)
state = {}
with torch.no_grad():
    for _ in range(3): hc.eval_entry(mods, cfg, name, fr, state)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        hc.eval_entry(mods, cfg, name, fr, state)
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.kernels or not e.name.startswith("aten::"): continue
    chain = []
    q = e.cpu_parent
    while q is not None and len(chain) < 4:
        chain.append(q.name); q = q.cpu_parent
    site = " < ".join(chain) if chain else "(top level)"
    if e.input_shapes: site += "  " + str(e.input_shapes)[:70]
    k = (e.name, site[:160])
    agg[k][0] += 1; agg[k][1] += sum(kk.duration for kk in e.kernels)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("total aten-launched kernel time: %.3f ms in %d launches" % (sum(v[1] for v in agg.values()) / 1e3, sum(v[0] for v in agg.values())))
for (n, s), (c, t) in rows[:45]: print(f"{n:16s} {c:3d} {t / 1e3:7.3f} ms  {s}")
