"""dev: which call sites launch the torch elementwise / copy / fill / cat / reduce kernels of one training iteration?
torch.profiler with stacks: per (kernel family, innermost selfocc_amd frame) the launch count and the device time."""
import sys, os, re, collections, runpy
sys.argv = [sys.argv[0]]
os.environ['SO_PROF_GLUE'] = '1'
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'bench_hotpath_train.py')).read()
src = src.split("encoder.train(); head.train()")[0]
g = {'__name__': 'bench', '__file__': os.path.join(here, 'bench_hotpath_train.py')}
exec(compile(src, 'bench_hotpath_train.py', 'exec'), g)
import torch
from torch.profiler import profile, ProfilerActivity
g['encoder'].train(); g['head'].train()
g['run'](2, False)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    g['run'](1, False)
torch.cuda.synchronize()
FAM = [('add', r'CUDAFunctor_add|CUDAFunctorOnSelf_add'), ('copy', r'copyBuffer|direct_copy|Memcpy'), ('fill', r'FillFunctor|fillBuffer|Memset'),
       ('cat', r'CatArray'), ('reduce', r'reduce_kernel'), ('dropout', r'dropout|masked_scale'), ('mul', r'MulFunctor|BinaryFunctor|AUnaryFunctor'),
       ('other_torch', r'at::native')]
def fam(name):
    for f, pat in FAM:
        if re.search(pat, name): return f
    return None
agg = collections.defaultdict(lambda: [0, 0.0])
ev = prof.events()
# kernels hang under their launching CPU op: walk cpu ops with a stack, sum device time of their kernels
for e in ev:
    if e.device_type.name != 'CPU' or not e.kernels: continue
    par, p = [], e.cpu_parent
    while p is not None and len(par) < 3:
        par.append(p.name[:34]); p = p.cpu_parent
    shapes = str([tuple(x) for x in (e.input_shapes or []) if x])[:70]
    site = shapes + ' <- ' + ' <- '.join(par)
    for k in e.kernels:
        f = fam(k.name)
        if f is None: continue
        a = agg[(f, e.name, site)]
        a[0] += 1; a[1] += k.duration
tot = collections.defaultdict(float)
for (f, op, site), (n, us) in agg.items(): tot[f] += us
print({k: round(v / 1e3, 2) for k, v in tot.items()}, 'ms per iteration')
for (f, op, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f"{us/1e3:7.3f} ms {n:4d} x {f:8s} {op[:40]:40s} {site[:170]}")
