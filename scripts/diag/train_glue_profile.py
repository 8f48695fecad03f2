"""Dev: which Python call sites launch the torch-side fill / add / copy kernels of one nuscenes_occ training iteration
(torch.profiler with stacks; the kernels' own time is in profiles/*_train_kernel_trace.txt)."""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "scripts"))
import numpy as np, torch
import hotpath_common as hc
from torch.profiler import profile, ProfilerActivity
d = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_occ"
os.environ["eval"] = "false"
torch.manual_seed(0); np.random.seed(0)
cfg = hc.shipped(name)
mods = hc.build(cfg, d, want_loss=True)
for m in mods[:3]: m.train()
params = [p for m in mods[:3] for p in m.parameters()]
opt_cfg = dict(cfg["optimizer"]["optimizer"]); opt_cfg.pop("type")
opt = torch.optim.AdamW(params, **opt_cfg)
fr = hc.frame_inputs(cfg, name, d, seed=0)
def step(it):
    opt.zero_grad(set_to_none=True)
    hc.train_iteration(mods, cfg, fr, global_iter=it)
    torch.nn.utils.clip_grad_norm_(params, cfg["grad_max_norm"]); opt.step()
for it in range(3): step(it)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(3)
    torch.cuda.synchronize()
WANT = ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::copy_", "aten::sum", "aten::mul", "aten::cat", "aten::clone", "aten::contiguous")
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name not in WANT or e.device_time_total <= 0: continue
    # only leaf-ish: events whose own kernels exist
    if not e.kernels: continue
    chain = []
    q = e.cpu_parent
    while q is not None and len(chain) < 4:
        chain.append(q.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
        q = q.cpu_parent
    site = " < ".join(chain) if chain else "(top level)"
    if e.input_shapes: site += "  " + str(e.input_shapes)[:60]
    k = (e.name, site[:150])
    agg[k][0] += 1; agg[k][1] += sum(kk.duration for kk in e.kernels)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = collections.defaultdict(lambda: [0, 0.0])
for (n, s), (c, t) in rows: tot[n][0] += c; tot[n][1] += t
print("per op:", {n: (c, round(t / 1e3, 3)) for n, (c, t) in tot.items()})
for (n, s), (c, t) in rows[:60]: print(f"{n:14s} {c:4d} {t / 1e3:8.3f} ms  {s}")
