"""GPU parity of the fused reprojection sampling (selfocc_reproj_fwd/_bwd) vs the torch-op
port of the reference loss lines (forward + autograd gradient wrt the weights)."""
import math

import numpy as np
import pytest
import torch

from oracle import torch_port as tp
from selfocc_amd.reproj import ReprojSampleFunction

pytestmark = pytest.mark.gpu
D0 = torch.device("cuda:0")


def make_case(R=300, S=64, Hi=96, Wi=200, seed=0, with_deltas=False):
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    f = 0.8 * Wi
    K = np.array([[f, 0, Wi / 2, 0], [0, f, Hi / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    def motion(yaw_deg, tx, tz):
        y = math.radians(yaw_deg)
        Rm = np.array([[math.cos(y), 0, math.sin(y), tx], [0, 1, 0, 0.02], [-math.sin(y), 0, math.cos(y), tz], [0, 0, 0, 1]])
        return torch.tensor(K @ Rm @ np.linalg.inv(K), dtype=torch.float32)
    T_prev, T_next = motion(2.0, 0.3, -0.6), motion(-2.5, -0.2, 0.7)
    img_prev, img_next = torch.rand(3, Hi, Wi, generator=g), torch.rand(3, Hi, Wi, generator=g)
    pix = torch.stack([torch.rand(R, generator=g) * Wi, torch.rand(R, generator=g) * Hi], -1)
    curr_rgb = torch.rand(R, 3, generator=g)
    near = torch.rand(R, 1, generator=g) * 0.5
    far = 2.0 + torch.rand(R, 1, generator=g) * 40.0
    far[: R // 20] = near[: R // 20] + 1e-6        # degenerate rays (near == far)
    edges = near + (far - near) * torch.linspace(0, 1, S + 1)[None]
    ts = ((edges[:, :-1] + edges[:, 1:]) / 2).contiguous()
    deltas = (edges[:, 1:] - edges[:, :-1]).contiguous() if with_deltas else None
    weights = torch.softmax(torch.randn(R, S, generator=g) * 3, -1) * torch.rand(R, 1, generator=g)
    weights[R // 2: R // 2 + 5] = 0.0               # rays with no weight at all
    return weights, ts, deltas, pix, curr_rgb, T_prev, T_next, img_prev, img_next, float(Hi), float(Wi)


@pytest.mark.parametrize("S,with_deltas", [(64, False), (32, True), (256, False), (100, True)])
def test_reproj_fwd_bwd_vs_port(hip, S, with_deltas):
    case = make_case(S=S, seed=S, with_deltas=with_deltas)
    w = case[0].clone().requires_grad_(True)
    l1, comb, anyv = tp.reproj_sample_port(w, *case[1:])
    g = torch.Generator().manual_seed(1)
    g1, g2 = torch.randn(l1.shape, generator=g), torch.randn(comb.shape, generator=g)
    ((l1 * g1).sum() + (comb * g2).sum()).backward()

    dev = [None if t is None else (t.to(D0) if torch.is_tensor(t) else t) for t in case]
    wd = dev[0].clone().requires_grad_(True)
    hl1, hcomb, hany = ReprojSampleFunction.apply(wd, *dev[1:])
    assert torch.equal(hany.cpu(), anyv)
    assert torch.allclose(hl1.cpu(), l1.detach(), rtol=1e-4, atol=1e-6)
    assert torch.allclose(hcomb.cpu(), comb.detach(), rtol=1e-4, atol=1e-6)
    ((hl1 * g1.to(D0)).sum() + (hcomb * g2.to(D0)).sum()).backward()
    assert torch.allclose(wd.grad.cpu(), w.grad, rtol=1e-3, atol=1e-4 * w.grad.abs().max().item())
    assert 0.2 < anyv.mean() <= 1.0


def test_reproj_all_invalid_and_empty(hip):
    case = list(make_case(R=40, S=16))
    case[5] = torch.diag(torch.tensor([1.0, 1.0, -1.0, 1.0]))      # projects behind both cameras
    case[6] = torch.diag(torch.tensor([1.0, 1.0, -1.0, 1.0]))
    dev = [None if t is None else (t.to(D0) if torch.is_tensor(t) else t) for t in case]
    l1, comb, anyv = ReprojSampleFunction.apply(*dev)
    assert anyv.sum() == 0 and l1.abs().max() == 0 and comb.abs().max() == 0
    empty = [None if t is None else (t[:0].to(D0) if (torch.is_tensor(t) and t.shape[0] == 40) else (t.to(D0) if torch.is_tensor(t) else t)) for t in case]
    l1, comb, anyv = ReprojSampleFunction.apply(*empty)
    assert l1.numel() == 0
