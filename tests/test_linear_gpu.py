"""selfocc_linear_wgrad (csrc/linear.hip): dW = dy^T x and db = colsum(dy) in one MFMA pass, vs float64 torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from selfocc_amd._lib import lib
    return lib()


@pytest.mark.parametrize("T,N,K", [(66049, 384, 96), (78899, 432, 96), (7967, 2304, 96), (78899, 96, 192),
                                   (8200, 25, 96), (10001, 216, 96), (9000, 96, 32), (8192, 70, 64), (8193, 33, 128),
                                   (70, 96, 96), (1, 5, 96)])
def test_linear_wgrad_matches_f64(hip, T, N, K):
    from selfocc_amd.linear import linear_wgrad, wgrad_supported
    assert wgrad_supported(T, N, K)
    g = torch.Generator().manual_seed(T + N)
    dy = torch.randn(T, N, generator=g).cuda()
    x = torch.randn(T, K, generator=g).cuda()
    dw, db = linear_wgrad(dy, x)
    want_w = (dy.double().t() @ x.double())
    want_b = dy.double().sum(0)
    scale = want_w.abs().max().item()
    assert (dw.double() - want_w).abs().max().item() < 2e-6 * max(scale, 1.0) * max(1.0, (T / 1000) ** 0.5)
    assert (db.double() - want_b).abs().max().item() < 2e-6 * max(want_b.abs().max().item(), 1.0) * max(1.0, (T / 1000) ** 0.5)
    # deterministic: fixed summation order
    dw2, db2 = linear_wgrad(dy, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dw3, none = linear_wgrad(dy, x, with_bias=False)
    assert none is None and torch.equal(dw, dw3)


def test_linear_wgrad_unsupported_shape_is_loud(hip):
    from selfocc_amd.linear import linear_wgrad, wgrad_supported
    assert not wgrad_supported(1000, 96, 100)
    with pytest.raises(RuntimeError):
        linear_wgrad(torch.randn(1000, 96).cuda(), torch.randn(1000, 100).cuda())


def test_tall_linear_backward_uses_fused_wgrad(hip):
    """TallLinear under autograd (fused wgrad) == nn.Linear under autograd."""
    from selfocc_amd.model import bricks
    torch.manual_seed(0)
    lin = bricks.TallLinear(96, 216).cuda()
    ref = torch.nn.Linear(96, 216).cuda()
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(1, 9000, 96).cuda()
    g = torch.randn(1, 9000, 216).cuda()
    xa = x.clone().requires_grad_(True); lin(xa).backward(g)
    xb = x.clone().requires_grad_(True); ref(xb).backward(g)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(lin.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(lin.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-3)
