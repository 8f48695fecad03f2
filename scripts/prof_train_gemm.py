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
for e in prof.events():
    if e.device_type.name!='CPU' or not e.kernels: continue
    for k in e.kernels:
        if k.name.startswith('Cijk_'):
            par=[];p=e.cpu_parent
            while p is not None and len(par)<4: par.append(p.name[:40]); p=p.cpu_parent
            print(f"{k.duration:8.1f} us {k.name[:60]} | {e.name} {[tuple(x) for x in (e.input_shapes or []) if x]} <- {' <- '.join(par)}")
