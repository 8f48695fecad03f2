"""Host side of selfocc_linear_wgrad: weight and bias gradient of a Linear with very many rows in one pass over
dy and x (csrc/linear.hip).  Used by model.bricks._TallLinear (the encoder's projections, the field MLP)."""
import torch

from ._lib import lib, check, ptr, current_stream


def wgrad_supported(rows, n_out, n_in):
    return lib().selfocc_linear_wgrad_supported(int(rows), int(n_out), int(n_in)) == 1


def linear_wgrad(dy, x, with_bias=True):
    """dy (T, N), x (T, K) float32 CUDA(HIP) contiguous -> (dw (N, K), db (N) | None)."""
    if not dy.is_cuda:
        raise RuntimeError("linear_wgrad needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
    T, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == T and dy.dtype == torch.float32 and x.dtype == torch.float32
    dy, x = dy.contiguous(), x.contiguous()
    dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)
    db = torch.empty(N, device=dy.device, dtype=torch.float32) if with_bias else None
    nbytes = int(lib().selfocc_linear_wgrad_workspace(T, N, K))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    check(lib().selfocc_linear_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db), T, N, K, ptr(ws), nbytes,
                                     current_stream(dy.device)), "selfocc_linear_wgrad")
    return dw, db


def dgrad_supported(rows, n_out, n_in):
    return lib().selfocc_linear_dgrad_supported(int(rows), int(n_out), int(n_in)) == 1


def linear_dgrad(dy, weight):
    """dx (T, K) = dy (T, N) @ weight (N, K), float32 CUDA(HIP) (csrc/linear_fwd.hip, linear_dgrad_b3_kernel)."""
    if not dy.is_cuda:
        raise RuntimeError("linear_dgrad needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
    T, N = dy.shape
    K = weight.shape[1]
    assert weight.shape[0] == N and dy.dtype == torch.float32 and weight.dtype == torch.float32
    dy, weight = dy.contiguous(), weight.contiguous()
    dx = torch.empty(T, K, device=dy.device, dtype=torch.float32)
    nbytes = int(lib().selfocc_linear_dgrad_workspace(N, K))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    check(lib().selfocc_linear_dgrad(ptr(dy), ptr(weight), ptr(dx), T, N, K, ptr(ws), nbytes, current_stream(dy.device)),
          "selfocc_linear_dgrad")
    return dx


def linear_fwd_supported(rows, n_out, n_in):
    return lib().selfocc_linear_fwd_supported(int(rows), int(n_out), int(n_in)) == 1


def linear_fwd(x, weight, bias=None, relu=False, residual=None, ln=None, out=None, want_stats=False):
    """y = LN?(relu?(x W^T + b) + residual) in one MFMA-f32 pass (csrc/linear_fwd.hip).

    x (T, K), weight (N, K), bias (N) | None, residual (T, N) | None (row stride free, unit column stride),
    ln = (gamma (N), beta (N), eps) | None, out (T, N) | None (row stride free: a column / row slice of a larger buffer).
    -> y, or (y, y_pre, mean, rstd) with ``want_stats`` (the LayerNorm input and statistics selfocc_layernorm_bwd needs)."""
    from .abi import LINEAR_RELU
    if not x.is_cuda:
        raise RuntimeError("linear_fwd needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
    T, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and x.dtype == torch.float32 and weight.dtype == torch.float32
    x, weight = x.contiguous(), weight.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    if out is None:
        out = torch.empty(T, N, device=x.device, dtype=torch.float32)
    assert out.shape == (T, N) and out.stride(1) == 1 and out.dtype == torch.float32
    ldr = 0
    if residual is not None:
        assert residual.shape == (T, N) and residual.dtype == torch.float32
        if residual.stride(1) != 1:
            residual = residual.contiguous()
        ldr = residual.stride(0) if T > 1 else N
    g = b = y_pre = mean = rstd = None
    eps = 0.0
    if ln is not None:
        g, b, eps = ln
        g, b = g.contiguous(), b.contiguous()
        if want_stats:
            y_pre = torch.empty(T, N, device=x.device, dtype=torch.float32)
            mean = torch.empty(T, device=x.device, dtype=torch.float32)
            rstd = torch.empty(T, device=x.device, dtype=torch.float32)
    elif want_stats:
        raise ValueError("want_stats needs ln")
    ldy = out.stride(0) if T > 1 else N
    check(lib().selfocc_linear_fwd(ptr(x), ptr(weight), ptr(bias), ptr(residual), int(ldr), ptr(g), ptr(b), float(eps),
                                   ptr(out), int(ldy), ptr(y_pre), ptr(mean), ptr(rstd), T, N, K,
                                   LINEAR_RELU if relu else 0, current_stream(x.device)), "selfocc_linear_fwd")
    return (out, y_pre, mean, rstd) if want_stats else out


def linear_fwd_heads_supported(rows, n_out, n_in, nv):
    return (n_out >= 96 and n_out % 96 == 0 and nv >= 16 and rows % nv == 0 and rows * n_out < 2 ** 31
            and linear_fwd_supported(rows, n_out, n_in))


def linear_fwd_heads(x, weight, bias, nv, relu=False):
    """value_proj with a head-major result (csrc/linear_fwd.hip, selfocc_linear_fwd_heads): x (B * nv, K), weight
    (G * 96, K) -> (G, B, 6, nv, 16) float32 — per group the (bs, heads, nv, d) tensor the fused / camera-loop MSDA ops
    take with ``head_major=True``."""
    from .abi import LINEAR_RELU
    if not x.is_cuda:
        raise RuntimeError("linear_fwd_heads needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
    T, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and x.dtype == torch.float32 and weight.dtype == torch.float32
    x, weight = x.contiguous(), weight.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    y = torch.empty(N // 96 if N % 96 == 0 else 0, T // nv if nv else 0, 6, nv, 16, device=x.device, dtype=torch.float32)
    check(lib().selfocc_linear_fwd_heads(ptr(x), ptr(weight), ptr(bias), ptr(y), T, N, K, int(nv),
                                         LINEAR_RELU if relu else 0, current_stream(x.device)), "selfocc_linear_fwd_heads")
    return y
