"""dev: HIP-event time of selfocc_field_volume_bwd at the nuscenes_occ size (257x257x25, 25 outputs); SELFOCC_HIP_LIB selects an A/B build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from selfocc_amd.field import FieldVolumeFunction
d = torch.device("cuda:0"); torch.manual_seed(0)
H, W, D, C, color = 257, 257, 25, 96, 24
hw, zh, wz = (torch.randn(n, C, device=d, requires_grad=True) for n in (H * W, D * H, W * D))
lins = [nn.Linear(C, C).to(d), nn.Linear(C, 1 + color).to(d)]
ts = []
for it in range(8):
    sdf, feat = FieldVolumeFunction.apply(hw, zh, wz, lins[0].weight, lins[0].bias, lins[1].weight, lins[1].bias, (H, W, D), color)
    gs, gf = torch.randn_like(sdf), torch.randn_like(feat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.autograd.backward([sdf, feat], [gs, gf])
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(os.environ.get("SELFOCC_HIP_LIB", "default").split("/")[-1], "field_volume backward ms", round(sorted(ts[2:])[3], 3), "g_hw sum", float(hw.grad.double().sum()))
