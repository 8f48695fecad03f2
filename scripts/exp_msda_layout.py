"""dev experiment: is the MSDA forward bound by L1 line fills (128-B lines, 64-B corner segments)?
(a) baseline (bs, nv, 6 heads, 16); (b) all points on one pixel (everything hits in L1); (c) head-major emulation:
the 6 heads as 6 extra batch items with 1 head each, so that horizontally adjacent pixels of a head are adjacent in memory."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd.msda import MultiScaleDeformableAttnFunction as F
d = torch.device("cuda:0"); torch.manual_seed(0)
shapes = [[96, 200], [48, 100], [24, 50], [12, 25]]
bs, nq, P, H, D = 6, 22016, 8, 6, 16
sh = torch.tensor(shapes, device=d); st = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
nv = int((sh[:, 0] * sh[:, 1]).sum()); L = 4
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
side = int(nq ** 0.5) + 1; qi = torch.arange(nq, device=d)
base = torch.stack([(qi % side) / side, (qi // side) / side], -1)
wh = torch.stack([sh[:, 1], sh[:, 0]], -1).float()
off = torch.randn(bs, nq, H, L, P, 2, device=d) * 2.0
loc = base[None, :, None, None, None, :] + off / wh[None, None, None, :, None, :]
attw = torch.softmax(torch.randn(bs, nq, H, L * P, device=d), -1).view(bs, nq, H, L, P)
value = torch.randn(bs, nv, H, D, device=d)
with torch.no_grad():
    t_a = timeit(lambda: F.apply(value, sh, st, loc, attw, 64))
    loc_same = torch.full_like(loc, 0.5)
    t_b = timeit(lambda: F.apply(value, sh, st, loc_same, attw, 64))
    v2 = value.permute(0, 2, 1, 3).reshape(bs * H, nv, 1, D).contiguous()
    loc2 = loc.permute(0, 2, 1, 3, 4, 5).reshape(bs * H, nq, 1, L, P, 2).contiguous()
    aw2 = attw.permute(0, 2, 1, 3, 4).reshape(bs * H, nq, 1, L, P).contiguous()
    t_c = timeit(lambda: F.apply(v2, sh, st, loc2, aw2, 64))
    o1 = F.apply(value, sh, st, loc, attw, 64).view(bs, nq, H, D)
    o2 = F.apply(v2, sh, st, loc2, aw2, 64).view(bs, H, nq, D).permute(0, 2, 1, 3)
    print("match", torch.allclose(o1, o2, atol=1e-5))
print(f"(a) baseline {t_a:.4f} ms   (b) all points on one pixel {t_b:.4f} ms   (c) head-major emulation {t_c:.4f} ms")
