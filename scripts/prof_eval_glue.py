"""dev: which ops launch the non-HIP-library kernels (copies, fills, RNG, cat, elementwise) of one depth-eval frame?"""
import sys, os, re, collections
here = os.path.dirname(os.path.abspath(__file__))
OCC = len(sys.argv) > 1 and sys.argv[1] == 'occ'          # `prof_eval_glue.py occ`: the occupancy-evaluation frame
name = 'bench_hotpath_occ.py' if OCC else 'bench_hotpath_eval.py'
src = open(os.path.join(here, name)).read().split("def ev():")[0]
g = {'__name__': 'bench', '__file__': os.path.join(here, name)}
sys.argv = sys.argv[:1]
exec(compile(src, name, 'exec'), g)
import torch
from torch.profiler import profile, ProfilerActivity
enc, lifter, head, feats, metas = g['encoder'], g['lifter'], g['head'], g['feats'], g['metas']
def frame():
    rep = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
    if OCC:
        res = head.forward_occ(rep, metas, aabb=g['pcr'], resolution=0.4)
        return g['occ_resample'](res['sdf'], g['g'], 0.0, logits=res['logits'], lut=g['OPENSEED2NUSCENES'], crop=(6, 6, 6, 6, 0, 4))
    head.prepare(rep, metas)
    return head.render(metas, batch=90000)
with torch.no_grad():
    frame(); frame(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        frame()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != 'CPU' or not e.kernels: continue
    par, p = [], e.cpu_parent
    while p is not None and len(par) < 3:
        par.append(p.name[:30]); p = p.cpu_parent
    shapes = str([tuple(x) for x in (e.input_shapes or []) if x])[:60]
    for k in e.kernels:
        if 'anonymous namespace)::' in k.name and 'at::native' not in k.name: continue     # our HIP kernels
        a = agg[(k.name[:50], e.name, shapes + ' <- ' + ' <- '.join(par))]
        a[0] += 1; a[1] += k.duration
print(f"total {sum(v[1] for v in agg.values())/1e3:.3f} ms")
for (kn, op, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{us/1e3:7.3f} ms {n:4d} x {kn:50s} {op[:28]:28s} {site[:120]}")
