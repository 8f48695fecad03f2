"""Diagnostic (GPU): capture the arguments of every distinct low-level op call of ONE shipped-size eval encoder pass, then replay
each call DIAG_REPEAT times and compare the result bitwise with its first replay.  Run two copies concurrently (or one copy
beside a foreign load) to see which op stops being repeatable when the GPU is shared."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
d = torch.device("cuda:0")
TAG = os.environ.get("DIAG_TAG", "")
if os.environ.get("DIAG_LOAD"):          # foreign load: plain torch GEMMs for DIAG_LOAD seconds
    a = torch.randn(8192, 8192, device=d)
    t0 = time.time()
    while time.time() - t0 < float(os.environ["DIAG_LOAD"]):
        for _ in range(20):
            a @ a
        torch.cuda.synchronize()
    sys.exit(0)
if os.environ.get("DIAG_SHIFT"):         # shift this process's virtual-address layout against its twin's
    _shift = torch.empty(int(float(os.environ["DIAG_SHIFT"]) * (1 << 30)), dtype=torch.uint8, device=d)
if os.environ.get("DIAG_TORCH"):         # control: a torch gather kernel (grid_sample) replayed the same way
    torch.manual_seed(1)
    img = torch.randn(6, 16, 116, 200, device=d)
    grid = torch.rand(6, 400, 400, 2, device=d) * 2.2 - 1.1
    first, bad = None, 0
    n = int(os.environ.get("DIAG_REPEAT", "200"))
    for it in range(n):
        o = torch.nn.functional.grid_sample(img, grid, align_corners=False)
        if first is None:
            first = o.clone()
        elif not torch.equal(first, o):
            bad += 1
    print(TAG, f"torch grid_sample non-repeatable {bad}/{n - 1}", flush=True)
    sys.exit(0)
import hotpath_common as hc
import selfocc_amd.model.bricks as bricks
import selfocc_amd.model.encoder.attention as attention
import selfocc_amd.model.encoder.tpvformer as tpvformer

def clone(x):
    if torch.is_tensor(x):
        return x.detach().clone()
    if isinstance(x, (list, tuple)):
        return type(x)(clone(y) for y in x)
    if isinstance(x, dict):
        return {k: clone(v) for k, v in x.items()}
    return x

def sig(x):
    if torch.is_tensor(x):
        return (tuple(x.shape), str(x.dtype), x.stride())
    if isinstance(x, (list, tuple)):
        return tuple(sig(y) for y in x)
    if isinstance(x, dict):
        return tuple((k, sig(v)) for k, v in sorted(x.items()))
    if isinstance(x, torch.nn.Module):
        return type(x).__name__
    return x if isinstance(x, (int, float, bool, str, type(None))) else type(x).__name__

calls = {}
def rec(name, fn):
    def w(*a, **k):
        key = (name, sig(a), sig(k))
        if key not in calls:
            calls[key] = (fn, a if name in KEEP_REF else clone(a), clone(k))     # non-contiguous views keep their strides via clone()?
        return fn(*a, **k)
    return w
KEEP_REF = set()
for mod in (bricks, attention, tpvformer):
    for name in ("linear_fwd", "linear_fwd_heads", "msda_fused_inference", "msda_cross_inference", "fused_linear",
                 "value_proj_head_major", "value_proj_head_major_multi", "point_sampling"):
        if hasattr(mod, name):
            setattr(mod, name, rec(f"{mod.__name__.split('.')[-1]}.{name}", getattr(mod, name)))

torch.manual_seed(0)
cfg = hc.shipped("nuscenes_occ")
lifter, enc, _h, _ = hc.build(cfg, d)
enc.eval()
img = tuple(cfg['img_size'])
c2w, l2i, K = hc.ring_cameras(6, img, 1266.0)
metas = [dict(lidar2img=l2i, img2lidar=c2w, img_shape=img)]
g = torch.Generator().manual_seed(5)
feats = [torch.randn(1, 6, 96, -(-img[0] // s_), -(-img[1] // s_), generator=g).to(d) for s_ in (8, 16, 32, 64)]
enc.layers = enc.layers[:1]
if os.environ.get("DIAG_ENCLOOP"):       # load twin: whole eval encoder passes for DIAG_ENCLOOP seconds, nothing checked
    t0 = time.time()
    with torch.no_grad():
        while time.time() - t0 < float(os.environ["DIAG_ENCLOOP"]):
            enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)
            torch.cuda.synchronize()
    sys.exit(0)
with torch.no_grad():
    enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)
    torch.cuda.synchronize()
    print(TAG, "captured", len(calls), "distinct calls", flush=True)
    n = int(os.environ.get("DIAG_REPEAT", "200"))
    def flat(x):
        if torch.is_tensor(x):
            return [x]
        if isinstance(x, (list, tuple)):
            return [t for y in x for t in flat(y)]
        if isinstance(x, dict):
            return [t for y in x.values() for t in flat(y)]
        return []
    only = os.environ.get("DIAG_ONLY", "")
    for (name, sa, sk), (fn, a, k) in calls.items():
        if only and only not in name:
            continue
        first, bad, worst = None, 0, 0.0
        for it in range(n):
            out = [t.clone() for t in flat(fn(*a, **k))]
            if first is None:
                first = out
                continue
            if it % 16 == 0:
                torch.cuda.synchronize()
            ne = [not torch.equal(x, y) for x, y in zip(first, out)]
            if any(ne):
                bad += 1
                if bad <= 3:
                    for x, y in zip(first, out):
                        df = (x != y).reshape(-1, 16) if x.shape[-1] % 16 == 0 else (x != y).reshape(-1, 1)
                        rows = df.any(-1).nonzero().flatten()
                        print(TAG, f"   replay {it}: {rows.numel()} (query, head) rows of {df.shape[0]} differ; channels per row "
                              f"{df[rows].sum(-1)[:12].tolist()}; rows {rows[:24].tolist()}", flush=True)
                        xr, yr = x.reshape(-1, df.shape[1])[rows[:3]], y.reshape(-1, df.shape[1])[rows[:3]]
                        print(TAG, "      first ", xr.tolist()[:2], flush=True)
                        print(TAG, "      replay", yr.tolist()[:2], flush=True)
                worst = max(worst, max(float((x.float() - y.float()).abs().max()) for x, y in zip(first, out)))
        shapes = [s[0] for s in sa if isinstance(s, tuple) and len(s) == 3 and isinstance(s[0], tuple)][:2]
        print(TAG, f"{name:40s} {str(shapes):60s} non-repeatable {bad}/{n - 1}  worst |diff| {worst:.3e}", flush=True)
