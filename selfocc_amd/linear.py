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
