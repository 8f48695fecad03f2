"""dev: MSDA forward + backward a few times at the nuscenes_occ hw-plane shape (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd.msda import MultiScaleDeformableAttnFunction as F
d = torch.device("cuda:0"); torch.manual_seed(0)
P = int(os.environ.get("P", 8)); nq = int(os.environ.get("NQ", 22016))
bs, shapes = 6, [[96, 200], [48, 100], [24, 50], [12, 25]]
sh = torch.tensor(shapes, device=d); st = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
nv = int((sh[:, 0] * sh[:, 1]).sum()); L = 4; H = 6; D = 16
value = torch.randn(bs, nv, H, D, device=d, requires_grad=True)
side = int(nq ** 0.5) + 1; qi = torch.arange(nq, device=d)
base = torch.stack([(qi % side) / side, (qi // side) / side], -1)
loc = (base[None, :, None, None, None, :] + torch.randn(bs, nq, H, L, P, 2, device=d) * 0.02).requires_grad_(True)
attw = torch.softmax(torch.randn(bs, nq, H, L * P, device=d), -1).view(bs, nq, H, L, P).requires_grad_(True)
g = None
for _ in range(4):
    out = F.apply(value, sh, st, loc, attw, 64)
    if g is None: g = torch.randn_like(out)
    out.backward(g)
torch.cuda.synchronize()
