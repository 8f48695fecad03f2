import sys, os, re, collections
sys.argv=[sys.argv[0]]
here=os.path.dirname(os.path.abspath(__file__))
src=open(os.path.join(here,'bench_hotpath_train.py')).read().split("encoder.train(); head.train()")[0]
g={'__name__':'bench','__file__':os.path.join(here,'bench_hotpath_train.py')}
exec(compile(src,'bench_hotpath_train.py','exec'),g)
import torch
from torch.profiler import profile, ProfilerActivity
g['encoder'].train(); g['head'].train()
g['run'](2,False)
with profile(activities=[ProfilerActivity.CPU,ProfilerActivity.CUDA],record_shapes=True) as prof:
    g['run'](1,False)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != 'CPU' or not e.kernels: continue
    for k in e.kernels:
        ours = '(anonymous namespace)::' in k.name and 'at::native' not in k.name
        if ours: continue
        par = []; p = e.cpu_parent
        while p is not None and len(par) < 3: par.append(p.name[:36]); p = p.cpu_parent
        key = (k.name[:44], e.name[:30], str([tuple(x) for x in (e.input_shapes or []) if x])[:70] + ' <- ' + ' <- '.join(par))
        agg[key][0] += 1; agg[key][1] += k.duration
print(f"non-selfocc kernels: {sum(v[1] for v in agg.values())/1e3:.2f} ms per iteration")
for (kn, op, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOPN', 40))]:
    print(f"{us/1e3:7.3f} ms {n:4d} x {kn:44s} {op:30s} {site[:150]}")
