"""dev: dense pair vs merged [offsets | logits] rows, forward and backward of the fused (cross-view self-attention shape) and
camera-loop (hw plane) ops — HIP-event times of the whole autograd call and of the forward alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from selfocc_amd.msda import (msda_fused_inference, msda_cross_inference, MSDAFusedFunction, MSDACrossFunction, to_head_major)
d0 = torch.device("cuda:0")
g = torch.Generator(device=d0).manual_seed(1)
heads, d = 6, 16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


for name in ("self", "hw", "zh"):
    if name == "self":
        cams, nq, L, P = 0, 78899, 3, 12
        shapes = torch.tensor([[257, 257], [25, 257], [257, 25]])
    elif name == "hw":
        cams, nq, L, P = 6, 66049, 4, 8
        shapes = torch.tensor([[96, 200], [48, 100], [24, 50], [12, 25]])
    else:
        cams, nq, L, P = 6, 6425, 4, 48
        shapes = torch.tensor([[96, 200], [48, 100], [24, 50], [12, 25]])
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    sh, st = shapes.to(d0), starts.to(d0)
    host = [int(v) for v in shapes.reshape(-1).tolist()]
    v_hm = to_head_major(torch.randn(max(cams, 1), nv, heads, d, device=d0, generator=g))
    off = torch.randn(nq, heads, L, P, 2, device=d0, generator=g) * 2
    lg = torch.randn(nq, heads, L * P, device=d0, generator=g)
    ol = torch.cat([off.reshape(nq, -1), lg.reshape(nq, -1)], -1).contiguous()
    gout = torch.randn(nq, heads * d, device=d0, generator=g)
    res = {}
    if cams:
        ref = torch.rand(cams, nq, P, 2, device=d0, generator=g) * 1.2 - 0.1
        vis = torch.rand(cams, nq, device=d0, generator=g) < 0.35
        with torch.no_grad():
            res['fwd_dense'] = timeit(lambda: msda_cross_inference(v_hm, sh, st, ref, vis, off, lg, True))
            res['fwd_merged'] = timeit(lambda: msda_cross_inference(v_hm, sh, st, ref, vis, ol, None, True, (L, P)))

        def run(merged):
            v = v_hm.detach().requires_grad_(True)
            if merged:
                x = ol.detach().requires_grad_(True)
                MSDACrossFunction.apply(v, sh, st, ref, vis, x, None, host, True, False, (L, P)).backward(gout)
            else:
                o, l_ = off.detach().requires_grad_(True), lg.detach().requires_grad_(True)
                MSDACrossFunction.apply(v, sh, st, ref, vis, o, l_, host, True, False).backward(gout)
    else:
        ref = torch.rand(1, nq, L, P, 2, device=d0, generator=g) * 1.1 - 0.05
        with torch.no_grad():
            res['fwd_dense'] = timeit(lambda: msda_fused_inference(v_hm, sh, st, ref, 2, off[None], lg[None], True))
            res['fwd_merged'] = timeit(lambda: msda_fused_inference(v_hm, sh, st, ref, 2, ol[None], None, True, (L, P)))

        def run(merged):
            v = v_hm.detach().requires_grad_(True)
            if merged:
                x = ol[None].detach().requires_grad_(True)
                MSDAFusedFunction.apply(v, sh, st, ref, 2, x, None, host, True, False, (L, P)).backward(gout[None])
            else:
                o, l_ = off[None].detach().requires_grad_(True), lg[None].detach().requires_grad_(True)
                MSDAFusedFunction.apply(v, sh, st, ref, 2, o, l_, host, True, False).backward(gout[None])
    res['fwd_bwd_dense'] = timeit(lambda: run(False))
    res['fwd_bwd_merged'] = timeit(lambda: run(True))
    print(name, res, flush=True)
