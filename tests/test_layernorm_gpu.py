"""GPU: selfocc_layernorm_fwd / _bwd (csrc/layernorm.hip, behind FastLayerNorm) against float64 torch
LayerNorm — the op the reference's encoder layers build through mmcv's build_norm_layer(dict(type='LN'))."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D0 = torch.device("cuda:0")


@pytest.mark.parametrize("rows,C", [(78899, 96), (1000, 32), (7, 128), (1, 4), (0, 96), (4099, 64)])
def test_layernorm_fwd_bwd_vs_float64(hip, rows, C):
    from selfocc_amd.model.bricks import FastLayerNorm
    g = torch.Generator().manual_seed(rows + C)
    ln = FastLayerNorm(C).to(D0)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        ln.bias.copy_(torch.randn(C, generator=g) * 0.1)
    x = (torch.randn(rows, C, generator=g) * 3.0 + 1.5).to(D0).requires_grad_(True)
    dy = torch.randn(rows, C, generator=g).to(D0)
    y = ln(x.view(1, rows, C)).view(rows, C)
    y.backward(dy)
    ref = torch.nn.LayerNorm(C).double()
    with torch.no_grad():
        ref.weight.copy_(ln.weight.double().cpu()); ref.bias.copy_(ln.bias.double().cpu())
    xr = x.detach().cpu().double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.cpu().double())
    assert torch.allclose(y.detach().cpu().double(), yr.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(x.grad.cpu().double(), xr.grad, rtol=1e-4, atol=1e-5)
    scale = max(1.0, float(ref.weight.grad.abs().max())) if rows else 1.0
    assert torch.allclose(ln.weight.grad.cpu().double(), ref.weight.grad, rtol=1e-4, atol=1e-4 * scale)
    assert torch.allclose(ln.bias.grad.cpu().double(), ref.bias.grad, rtol=1e-4, atol=1e-4 * scale)


def test_layernorm_no_grad_and_fallbacks(hip):
    from selfocc_amd.model.bricks import FastLayerNorm
    ln = FastLayerNorm(96).to(D0)
    x = torch.randn(2, 50, 96, device=D0)
    with torch.no_grad():
        a = ln(x)
        b = torch.nn.functional.layer_norm(x, (96,), ln.weight, ln.bias, ln.eps)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    odd = FastLayerNorm(30).to(D0)                      # C % 4 != 0 -> nn.LayerNorm path
    assert odd(torch.randn(5, 30, device=D0)).shape == (5, 30)
    assert FastLayerNorm(96)(torch.randn(3, 96)).shape == (3, 96)   # CPU tensors: host-side tests keep working
