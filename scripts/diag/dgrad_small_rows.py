"""dev: dx = dy W at the MERGED round-6 shapes incl. the small-row zh / wz ones (6 425 x 3 456) that go to the vendor GEMM today
(bricks.DGRAD_MIN_ROWS = 32 768): selfocc_linear_dgrad vs `dy @ w`; and dW = dy^T x (selfocc_linear_wgrad vs torch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from selfocc_amd.linear import linear_dgrad, linear_wgrad, dgrad_supported, wgrad_supported
d = torch.device("cuda:0"); torch.manual_seed(0)


def t(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (T, N, K) in [(6425, 3456, 96), (7967, 3456, 96), (66049, 576, 96), (78899, 648, 96), (6425, 96, 96), (6425, 192, 96), (153000, 288, 96)]:
    dy = torch.randn(T, N, device=d); w = torch.randn(N, K, device=d); x = torch.randn(T, K, device=d)
    ours = t(lambda: linear_dgrad(dy, w)) if dgrad_supported(T, N, K) else float('nan')
    vend = t(lambda: dy @ w)
    wo = t(lambda: linear_wgrad(dy, x)) if wgrad_supported(T, N, K) else float('nan')
    wv = t(lambda: (dy.t() @ x, dy.sum(0)))
    print(f"T={T:6d} N={N:4d} K={K}: dgrad ours {ours:6.1f} us vendor {vend:6.1f} us | wgrad ours {wo:6.1f} us torch {wv:6.1f} us | dy stream @5TB/s {T * N * 4 / 5e6:5.1f} us")
