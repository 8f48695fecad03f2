"""dev: fused field-volume forward (inference) at the nuscenes sizes; SELFOCC_FIELD_B3=0/1 A/B across processes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from selfocc_amd.field import field_volume
d = torch.device("cuda:0"); torch.manual_seed(0)
for (H, W, D, color, F, dt) in [(257, 257, 25, 0, 0, torch.float32), (257, 257, 25, 24, 24, torch.float32),
                                 (257, 257, 25, 24, 24, torch.bfloat16), (200, 200, 16, 3, 4, torch.float32)]:
    C = 96
    hw, zh, wz = (torch.randn(n, C, device=d) for n in (H * W, D * H, W * D))
    lins = [nn.Linear(C, C).to(d), nn.Linear(C, 1 + color).to(d)]
    with torch.no_grad():
        for _ in range(5):
            field_volume(hw, zh, wz, (H, W, D), lins, F, dt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            field_volume(hw, zh, wz, (H, W, D), lins, F, dt)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    M = H * W * D
    fl = 2.0 * M * C * (C + 1 + color)
    print(f"B3={os.environ.get('SELFOCC_FIELD_B3','1')} {H}x{W}x{D} out={1+color} {str(dt)[6:]}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")
