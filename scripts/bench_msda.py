"""MSDA micro-benchmark at the reference's nuscenes_occ shapes:
cross-attention of the three TPV planes (6 cameras as batch, 25500 keys, 6 heads x 16, 4 levels,
P = 8 / 48 / 48) and the cross-view self-attention (78899 queries, 3 levels, P = 12).
Prints per-kernel time and ALGORITHMIC GB/s (value + loc + attw + out, each once)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd.msda import MultiScaleDeformableAttnFunction as F

d = torch.device("cuda:0")
torch.manual_seed(0)
CASES = {
    # name: (bs, nq, shapes, P, visible fraction of queries per camera)
    "cross_hw": (6, 66049 // 3, [[96, 200], [48, 100], [24, 50], [12, 25]], 8),
    "cross_zh": (6, 6425 // 3 * 2, [[96, 200], [48, 100], [24, 50], [12, 25]], 48),
    "self_xview": (1, 78899, [[257, 257], [25, 257], [257, 25]], 12),
}
res = {}
for name, (bs, nq, shapes, P) in CASES.items():
    sh = torch.tensor(shapes, device=d)
    st = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
    nv = int((sh[:, 0] * sh[:, 1]).sum()); L = len(shapes); H = 6; D = 16
    value = torch.randn(bs, nv, H, D, device=d)
    # projected-like locality: neighbouring queries -> neighbouring pixels, + N(0, 2px) offsets
    side = int(nq ** 0.5) + 1
    qi = torch.arange(nq, device=d)
    base = torch.stack([(qi % side) / side, (qi // side) / side], -1)            # nq, 2
    loc = base[None, :, None, None, None, :] + torch.randn(bs, nq, H, L, P, 2, device=d) * (2.0 / 100)
    attw = torch.softmax(torch.randn(bs, nq, H, L * P, device=d), -1).view(bs, nq, H, L, P)
    value.requires_grad_(True); loc.requires_grad_(True); attw.requires_grad_(True)
    out = F.apply(value, sh, st, loc, attw, 64)
    g = torch.randn_like(out)
    out.backward(g)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 10
    e[0].record()
    for _ in range(n):
        with torch.no_grad():
            F.apply(value, sh, st, loc, attw, 64)
    e[1].record()
    torch.cuda.synchronize()
    fwd_ms = e[0].elapsed_time(e[1]) / n
    tb = 0.0
    for _ in range(n):
        value.grad = loc.grad = attw.grad = None
        out = F.apply(value, sh, st, loc, attw, 64)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out.backward(g); b.record(); torch.cuda.synchronize()
        tb += a.elapsed_time(b)
    bwd_ms = tb / n
    pts = bs * nq * H * L * P
    alg_f = 4 * (value.numel() + loc.numel() + attw.numel() + out.numel())
    alg_b = 4 * (value.numel() * 2 + loc.numel() * 2 + attw.numel() * 2 + out.numel())
    res[name] = dict(points=pts, fwd_ms=round(fwd_ms, 4), bwd_ms_incl_memset=round(bwd_ms, 4),
                     fwd_alg_GBps=round(alg_f / fwd_ms / 1e6, 1), bwd_alg_GBps=round(alg_b / bwd_ms / 1e6, 1),
                     fwd_Gpts_per_s=round(pts / fwd_ms / 1e6, 2), alg_fwd_MB=round(alg_f / 1e6, 1))
    print(name, json.dumps(res[name]), flush=True)
