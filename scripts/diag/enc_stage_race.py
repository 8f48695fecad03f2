"""Diagnostic (GPU): which module of the shipped-size eval encoder is not bitwise repeatable when ANOTHER process shares the GPU?
Forward hooks hash every module call's inputs and outputs; for every later pass, the first call (execution order) whose input
hashes equal pass 0's but whose output hash differs is printed.  Run two copies concurrently."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import hotpath_common as hc
d = torch.device("cuda:0")
TAG = os.environ.get("DIAG_TAG", "")
torch.manual_seed(0)
cfg = hc.shipped("nuscenes_occ")
lifter, enc, _h, _ = hc.build(cfg, d)
enc.eval()
img = tuple(cfg['img_size'])
c2w, l2i, K = hc.ring_cameras(6, img, 1266.0)
metas = [dict(lidar2img=l2i, img2lidar=c2w, img_shape=img)]
g = torch.Generator().manual_seed(5)
feats = [torch.randn(1, 6, 96, -(-img[0] // s_), -(-img[1] // s_), generator=g).to(d) for s_ in (8, 16, 32, 64)]
enc.layers = enc.layers[:int(os.environ.get("DIAG_LAYERS", "4"))]

def tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            yield from tensors(y)
    elif isinstance(x, dict):
        for y in x.values():
            yield from tensors(y)

def h(x):
    hs = []
    for t in tensors(x):
        if t.is_cuda and t.numel():
            tt = t.detach().contiguous()
            if tt.dtype in (torch.float32, torch.int32):
                hs.append(tt.view(torch.int32).to(torch.int64).sum())
            elif tt.dtype == torch.bfloat16:
                hs.append(tt.view(torch.int16).to(torch.int64).sum())
            else:
                hs.append(tt.to(torch.float64).sum().view(torch.int64))
    return torch.stack(hs) if hs else torch.zeros(1, dtype=torch.int64, device=d)

trace = []
def hook(name):
    def f(mod, args, kwargs, out):
        trace.append((name, type(mod).__name__, h((args, kwargs)), h(out)))
    return f
for n, m in list(enc.named_modules()) + [("lifter." + n, m) for n, m in lifter.named_modules()]:
    m.register_forward_hook(hook(n), with_kwargs=True)

ref = None
n = int(os.environ.get("DIAG_REPEAT", "20"))
seen = {}
with torch.no_grad():
    for it in range(n):
        trace.clear()
        enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)
        torch.cuda.synchronize()
        cur = [(a, b, c.cpu(), e.cpu()) for a, b, c, e in trace]
        if ref is None:
            ref = cur
            continue
        for (na, ty, i0, o0), (nb, _, i1, o1) in zip(ref, cur):
            if i0.shape == i1.shape and torch.equal(i0, i1) and not torch.equal(o0, o1):
                seen[(na, ty)] = seen.get((na, ty), 0) + 1
                break
print(TAG, "calls per pass", len(ref), "first-divergent module counts over", n - 1, "passes:", flush=True)
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(TAG, "   ", v, k, flush=True)
