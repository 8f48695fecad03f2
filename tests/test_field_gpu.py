"""GPU parity: selfocc_field_volume_fwd (tri-plane sum + [Softplus, Linear] x n in one MFMA-f32 kernel) vs the
torch restatement of the same formula (float64 on CPU; neus_head.py:295-306 / bev_nerf.py:74-95)."""
import pytest
import torch
import torch.nn as nn

from selfocc_amd.field import field_volume

pytestmark = pytest.mark.gpu


def reference(hw, zh, wz, size, linears, dtype=torch.float64):
    H, W, D = size
    C = hw.shape[-1]
    x = hw.to(dtype).reshape(H, W, 1, C) + zh.to(dtype).reshape(D, H, 1, C).permute(1, 2, 0, 3) + \
        wz.to(dtype).reshape(W, D, 1, C).permute(2, 0, 1, 3)
    for lin in linears:
        x = torch.nn.functional.linear(torch.nn.functional.softplus(x), lin.weight.to(dtype), lin.bias.to(dtype))
    return x


CASES = [
    # H, W, D, C, n_linear, color_dims, feat_stride
    (9, 7, 5, 96, 2, 0, 0),        # depth config: sdf only; 315 rows = 9 full tiles + a ragged one
    (8, 8, 4, 96, 2, 24, 24),      # occ config: 3 rgb + 21 semantic channels
    (6, 5, 3, 96, 2, 3, 4),        # rgb only: feature volume padded to 4 channels
    (5, 6, 7, 64, 2, 5, 8),
    (4, 5, 6, 128, 2, 24, 24),
    (7, 3, 5, 96, 1, 24, 24),      # density_layers = 1: no hidden layer
    (1, 1, 1, 96, 2, 0, 0),        # a single voxel
]


@pytest.mark.parametrize("H,W,D,C,n_lin,color,F", CASES)
def test_field_volume_vs_torch(hip, H, W, D, C, n_lin, color, F):
    g = torch.Generator().manual_seed(H * 100 + W * 10 + D + C)
    hw, zh, wz = (torch.randn(n, C, generator=g) * 1.5 for n in (H * W, D * H, W * D))
    # exercise every Softplus regime: large negative (series), mid (log), > 20 (identity)
    hw[0, :8] = torch.tensor([-30.0, -12.0, -4.0, -1.0, 0.0, 6.0, 19.0, 25.0])
    lins = [nn.Linear(C, C) for _ in range(n_lin - 1)] + [nn.Linear(C, 1 + color)]
    for l in lins:
        nn.init.normal_(l.weight, std=0.3, generator=g)
        nn.init.normal_(l.bias, std=0.5, generator=g)
    ref = reference(hw, zh, wz, (H, W, D), lins)
    d = torch.device("cuda:0")
    lins_d = [l.to(d) for l in lins]
    sdf, feat = field_volume(hw.to(d), zh.to(d), wz.to(d), (H, W, D), lins_d, F)
    # tolerance stated for float paths: 1e-4 relative (BASELINE north_star); measured ~2e-6 here
    scale = ref.abs().max().item()
    assert torch.allclose(sdf.cpu().double(), ref[..., 0], rtol=1e-4, atol=1e-5 * scale)
    if F:
        assert feat.shape == (H, W, D, F)
        assert torch.allclose(feat[..., :color].cpu().double(), ref[..., 1:], rtol=1e-4, atol=1e-5 * scale)
        assert torch.all(feat[..., color:] == 0)
    else:
        assert feat is None
    # bf16 feature volume: same values rounded once
    if F:
        _, fb = field_volume(hw.to(d), zh.to(d), wz.to(d), (H, W, D), lins_d, F, torch.bfloat16)
        assert fb.dtype == torch.bfloat16
        assert torch.equal(fb, feat.to(torch.bfloat16))


def test_sdffield_fused_equals_unfused(hip):
    """SDFField.pre_compute_density_color: the inference (fused) path == the autograd (torch) path at the
    nuscenes_occ plane sizes scaled down 4x; the fused result feeds the same SDFVolume layout."""
    from selfocc_amd.model.head.neus_head import SDFField
    mapping_args = dict(nonlinear_mode='linear', h_size=[32, 0], h_range=[40.0, 0], h_half=False, w_size=[32, 0],
                        w_range=[40.0, 0], w_half=False, d_size=[12, 0], d_range=[-1.0, 5.4, 5.4])
    d = torch.device("cuda:0")
    torch.manual_seed(3)
    f = SDFField(mapping_args, embed_dims=96, color_dims=24, density_layers=2, sh_deg=0, tpv=True, return_sem=True).to(d)
    H, W, D = f.size_h, f.size_w, f.size_d
    rep = [torch.randn(1, H * W, 96, device=d), torch.randn(1, D * H, 96, device=d), torch.randn(1, W * D, 96, device=d)]
    with torch.no_grad():
        fused = f.pre_compute_density_color(rep)
        f.fused_volume = False
        plain = f.pre_compute_density_color(rep)
    assert fused.sdf.shape == plain.sdf.shape and fused.feat.shape == plain.feat.shape
    assert torch.allclose(fused.sdf, plain.sdf, rtol=1e-4, atol=1e-5)
    assert torch.allclose(fused.feat, plain.feat, rtol=1e-4, atol=1e-5)
    f.fused_volume = True
    # grad enabled: fused forward + fused backward (FieldVolumeFunction) vs the op-by-op autograd path
    grads = {}
    for fused in (True, False):
        f.fused_volume = fused
        f.zero_grad()
        reps = [r.clone().requires_grad_(True) for r in rep]
        vol = f.pre_compute_density_color(reps)
        assert vol.sdf.requires_grad
        assert (type(vol.sdf.grad_fn).__name__ == 'FieldVolumeFunctionBackward') == fused
        (vol.sdf.square().mean() + vol.feat.square().mean()).backward()
        grads[fused] = [r.grad.clone() for r in reps] + [p.grad.clone() for p in f.density_net.parameters()]
    f.fused_volume = True
    for a, b in zip(grads[True], grads[False]):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4 * b.abs().max().item())


def test_field_volume_rejects_bad_config(hip):
    from selfocc_amd._lib import SelfOccHipError
    d = torch.device("cuda:0")
    with pytest.raises(ValueError):
        field_volume(torch.zeros(4, 80, device=d), torch.zeros(4, 80, device=d), torch.zeros(4, 80, device=d), (2, 2, 2),
                     [nn.Linear(80, 3).to(d)], 0)
    with pytest.raises(RuntimeError):
        field_volume(torch.zeros(4, 96), torch.zeros(4, 96), torch.zeros(4, 96), (2, 2, 2), [nn.Linear(96, 1)], 0)


@pytest.mark.parametrize("H,W,D,color,F", [(9, 7, 25, 24, 24), (5, 6, 31, 0, 0), (4, 5, 11, 3, 4), (3, 70, 16, 24, 24), (2, 4, 40, 1, 4),
                                          (8, 8, 2, 3, 4), (1, 1, 1, 0, 0), (5, 3, 3, 24, 24)])     # whole / ragged 4 x 4 x 2 patches
def test_field_volume_backward_vs_autograd(hip, H, W, D, color, F):
    """FieldVolumeFunction (fused forward + fused backward) vs torch autograd of the same formula in float64:
    gradients of the three planes, both weights and both biases."""
    from selfocc_amd.field import FieldVolumeFunction
    C = 96
    g = torch.Generator().manual_seed(H * 100 + W * 10 + D)
    hw, zh, wz = (torch.randn(n, C, generator=g) * 1.2 for n in (H * W, D * H, W * D))
    hw[0, :8] = torch.tensor([-30.0, -12.0, -4.0, -1.0, 0.0, 6.0, 19.0, 25.0])
    l1, l2 = nn.Linear(C, C), nn.Linear(C, 1 + color)
    for l in (l1, l2):
        nn.init.normal_(l.weight, std=0.3, generator=g); nn.init.normal_(l.bias, std=0.5, generator=g)
    gs = torch.randn(H, W, D, generator=g)
    gf = torch.randn(H, W, D, F, generator=g) if F else None
    # float64 reference
    ins = [t.double().requires_grad_(True) for t in (hw, zh, wz, l1.weight.detach(), l1.bias.detach(), l2.weight.detach(), l2.bias.detach())]
    x = ins[0].reshape(H, W, 1, C) + ins[1].reshape(D, H, 1, C).permute(1, 2, 0, 3) + ins[2].reshape(W, D, 1, C).permute(2, 0, 1, 3)
    y = torch.nn.functional.linear(torch.nn.functional.softplus(x), ins[3], ins[4])
    o = torch.nn.functional.linear(torch.nn.functional.softplus(y), ins[5], ins[6])
    loss = (o[..., 0] * gs.double()).sum()
    if F:
        loss = loss + (o[..., 1:] * gf[..., :color].double()).sum()
    loss.backward()
    d = torch.device("cuda:0")
    dev = [t.detach().clone().to(d).requires_grad_(True) for t in (hw, zh, wz, l1.weight, l1.bias, l2.weight, l2.bias)]
    sdf, feat = FieldVolumeFunction.apply(*dev, (H, W, D), F)
    assert torch.allclose(sdf.detach().cpu().double(), o[..., 0].detach(), rtol=1e-4, atol=1e-4)
    lo = (sdf * gs.to(d)).sum()
    if F:
        lo = lo + (feat * gf.to(d)).sum()
    lo.backward()
    names = ["hw", "zh", "wz", "w1", "b1", "w2", "b2"]
    for n, a, b in zip(names, dev, ins):
        ref = b.grad
        scale = ref.abs().max().item()
        # float32 sums over up to H*W*D rows: 1e-4 of each tensor's scale (measured ~1e-6)
        assert torch.allclose(a.grad.cpu().double(), ref, rtol=1e-3, atol=1e-4 * scale), (n, (a.grad.cpu().double() - ref).abs().max().item(), scale)
