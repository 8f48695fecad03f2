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


# ---- selfocc_linear_fwd (csrc/linear_fwd.hip): y = LN?(relu?(x W^T + b) + residual) -------------------------------
FWD_SHAPES = [(66049, 384, 96), (78899, 432, 96), (7967, 2304, 96), (78899, 96, 192), (78899, 216, 96), (8200, 25, 96),
              (9000, 96, 32), (8192, 70, 64), (8193, 33, 128), (70, 96, 96), (1, 5, 96), (129, 288, 96), (4099, 192, 96),
              (78899, 648, 96), (4099, 648, 96), (4101, 84, 96), (4101, 172, 64)]


@pytest.mark.parametrize("T,N,K", FWD_SHAPES)
def test_linear_fwd_matches_f64(hip, T, N, K):
    """float32 MFMA = exact fmaf chains: within a few ulp of the float64 product; bias / ReLU / residual epilogues;
    strided output (a column block of a wider buffer) and strided residual."""
    from selfocc_amd.linear import linear_fwd, linear_fwd_supported
    assert linear_fwd_supported(T, N, K)
    g = torch.Generator().manual_seed(T + N + K)
    x = torch.randn(T, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    want = x.double() @ w.double().t() + b.double()
    y = linear_fwd(x, w, b)
    tol = 2e-6 * max(1.0, want.abs().max().item())
    assert (y.double() - want).abs().max().item() < tol
    assert (linear_fwd(x, w, None).double() - (want - b.double())).abs().max().item() < tol
    assert torch.equal(linear_fwd(x, w, b), y)          # deterministic
    y = linear_fwd(x, w, b, relu=True)
    assert (y.double() - want.clamp_min(0)).abs().max().item() < tol
    wide = torch.randn(T, N + 40, generator=g).cuda()
    res = wide[:, 7:7 + N]                              # row stride N + 40
    buf = torch.full((T, 2 * N + 3), 7.0).cuda()
    out = buf[:, N:2 * N]
    r = linear_fwd(x, w, b, relu=True, residual=res, out=out)
    assert r.data_ptr() == out.data_ptr()
    assert (out.double() - (want.clamp_min(0) + res.double())).abs().max().item() < 2 * tol
    assert torch.all(buf[:, :N] == 7.0) and torch.all(buf[:, 2 * N:] == 7.0)      # nothing written outside the block
    if N % 4 == 0:            # the same with 16-byte aligned row starts (the float4 epilogue of the b3 kernel)
        wide = torch.randn(T, N + 40, generator=g).cuda()
        res = wide[:, 8:8 + N]
        buf = torch.full((T, 2 * N + 8), 7.0).cuda()
        out = buf[:, N:2 * N]
        linear_fwd(x, w, b, relu=True, residual=res, out=out)
        assert (out.double() - (want.clamp_min(0) + res.double())).abs().max().item() < 2 * tol
        assert torch.all(buf[:, :N] == 7.0) and torch.all(buf[:, 2 * N:] == 7.0)


@pytest.mark.parametrize("T,N,K", [(78899, 96, 96), (78899, 96, 192), (7967, 96, 96), (130, 96, 96), (4100, 64, 96),
                                   (4100, 40, 64), (33, 96, 128)])
def test_linear_fwd_layernorm_epilogue(hip, T, N, K):
    """output_proj + residual + norm of a TPVFormerLayer step in one launch == torch float64; the saved statistics are
    what selfocc_layernorm_bwd takes."""
    from selfocc_amd.linear import linear_fwd
    g = torch.Generator().manual_seed(T * 3 + N)
    x = torch.randn(T, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(T, N, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    beta = (0.1 * torch.randn(N, generator=g)).cuda()
    pre = x.double() @ w.double().t() + b.double() + res.double()
    want = torch.nn.functional.layer_norm(pre, (N,), gamma.double(), beta.double(), 1e-5)
    y, y_pre, mean, rstd = linear_fwd(x, w, b, residual=res, ln=(gamma, beta, 1e-5), want_stats=True)
    assert (y_pre.double() - pre).abs().max().item() < 4e-6 * max(1.0, pre.abs().max().item())
    assert (y.double() - want).abs().max().item() < 2e-5
    assert (mean.double() - pre.mean(1)).abs().max().item() < 1e-5
    assert (rstd.double() - 1 / (pre.var(1, unbiased=False) + 1e-5).sqrt()).abs().max().item() < 1e-4
    y2 = linear_fwd(x, w, b, residual=res, ln=(gamma, beta, 1e-5))
    assert torch.equal(y, y2)


def test_linear_fwd_unsupported_shape_is_loud(hip):
    from selfocc_amd.linear import linear_fwd, linear_fwd_supported
    assert not linear_fwd_supported(1000, 96, 100)
    with pytest.raises(RuntimeError):
        linear_fwd(torch.randn(1000, 100).cuda(), torch.randn(96, 100).cuda())
    with pytest.raises(RuntimeError):       # LayerNorm epilogue needs the whole row in one column block
        linear_fwd(torch.randn(1000, 96).cuda(), torch.randn(192, 96).cuda(), ln=(torch.ones(192).cuda(), torch.zeros(192).cuda(), 1e-5))


@pytest.mark.parametrize("B,nv,G,K", [(6, 25500, 3, 96), (1, 78899, 1, 96), (2, 4100, 2, 96), (3, 17, 1, 96), (2, 1000, 1, 192)])
def test_linear_fwd_heads_is_the_head_major_projection(hip, B, nv, G, K):
    """selfocc_linear_fwd_heads == selfocc_linear_fwd followed by the (b, pix, g, h, c) -> (g, b, h, pix, c) transposition,
    bit for bit (the same MFMA chain; only the store addresses differ)."""
    from selfocc_amd.linear import linear_fwd, linear_fwd_heads, linear_fwd_heads_supported
    T, N = B * nv, 96 * G
    assert linear_fwd_heads_supported(T, N, K, nv)
    g = torch.Generator().manual_seed(B * nv + G)
    x = torch.randn(T, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    y = linear_fwd(x, w, b)
    want = y.view(B, nv, G, 6, 16).permute(2, 0, 3, 1, 4).contiguous()
    got = linear_fwd_heads(x, w, b, nv)
    assert got.shape == (G, B, 6, nv, 16) and torch.equal(got, want)
    assert torch.equal(linear_fwd_heads(x, w, b, nv, relu=True), want.clamp_min(0))
    with pytest.raises(RuntimeError):
        linear_fwd_heads(x, w[:95], b[:95], nv)            # N must be whole 96-column groups


# ---- selfocc_linear_dgrad (csrc/linear_fwd.hip): dx = dy W, the reduction over the layer's outputs looped in 96-wide chunks ----
# (T, N = reduced, K = columns of dx): the training encoder's shapes (offsets / weights / value / output projections, FFN)
DGRAD_SHAPES = [(66049, 384, 96), (66049, 192, 96), (6425, 768, 96), (78899, 96, 96), (78899, 48, 96), (78899, 192, 96),
                (78899, 96, 192), (4099, 1152, 96), (8193, 200, 96), (70, 8, 96), (1, 96, 96), (33, 104, 288), (257, 2304, 384)]


@pytest.mark.parametrize("T,N,K", DGRAD_SHAPES)
def test_linear_dgrad_matches_f64(hip, T, N, K):
    from selfocc_amd.linear import linear_dgrad, dgrad_supported
    assert dgrad_supported(T, N, K)
    g = torch.Generator().manual_seed(T + N + K)
    dy = torch.randn(T, N, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * 0.3).cuda()
    dx = linear_dgrad(dy, w)
    want = dy.double() @ w.double()
    scale = want.abs().max().item()
    # float32-level accuracy from the exact three-way bfloat16 split (the vendor f32 GEMM measures the same 1e-6 here)
    assert (dx.double() - want).abs().max().item() < 3e-6 * max(scale, 1.0)
    assert torch.equal(dx, linear_dgrad(dy, w))


def test_linear_dgrad_unsupported_shape_is_loud(hip):
    from selfocc_amd.linear import linear_dgrad, dgrad_supported
    assert not dgrad_supported(1000, 100, 96) and not dgrad_supported(1000, 96, 100)
    with pytest.raises(RuntimeError):
        linear_dgrad(torch.randn(1000, 96).cuda(), torch.randn(96, 100).cuda())


def test_tall_linear_backward_uses_fused_dgrad(hip, monkeypatch):
    from selfocc_amd.model import bricks
    import selfocc_amd.linear as L
    calls = []
    real = L.linear_dgrad
    monkeypatch.setattr(bricks, "linear_dgrad", lambda dy, w: (calls.append(dy.shape), real(dy, w))[1])
    torch.manual_seed(0)
    lin = bricks.TallLinear(96, 384).cuda()
    T = bricks.DGRAD_MIN_ROWS + 77
    x = torch.randn(1, T, 96).cuda().requires_grad_(True)
    g = torch.randn(1, T, 384).cuda()
    lin(x).backward(g)
    assert calls == [torch.Size([T, 384])]
    want = g.double().reshape(T, 384) @ lin.weight.double()
    assert (x.grad.double().reshape(T, 96) - want).abs().max().item() < 3e-6 * want.abs().max().item()
