"""dev: fused field-volume forward + backward at the nuscenes_occ size (257 x 257 x 25, 25 outputs), for rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from selfocc_amd.field import FieldVolumeFunction
d = torch.device("cuda:0"); torch.manual_seed(0)
H = W = 257; D = 25; C = 96; F = 24
hw, zh, wz = (torch.randn(n, C, device=d, requires_grad=True) for n in (H * W, D * H, W * D))
l1, l2 = nn.Linear(C, C).to(d), nn.Linear(C, 25).to(d)
for _ in range(5):
    sdf, feat = FieldVolumeFunction.apply(hw, zh, wz, l1.weight, l1.bias, l2.weight, l2.bias, (H, W, D), F)
    (sdf.mean() + feat.square().mean()).backward()
torch.cuda.synchronize(); print("ok")
