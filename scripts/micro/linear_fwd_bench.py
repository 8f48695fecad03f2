"""dev: selfocc_linear_fwd vs torch.addmm (hipBLASLt) on the encoder's Linear shapes (nuscenes_depth, dim 96)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from selfocc_amd.linear import linear_fwd
d = torch.device("cuda:0")
shapes = [("self off", 78899, 96, 432), ("self aw", 78899, 96, 216), ("self val/out", 78899, 96, 96),
          ("hw off", 66049, 96, 384), ("hw aw", 66049, 96, 192), ("hw out", 66049, 96, 96),
          ("zh off", 7967, 96, 2304), ("zh aw", 7967, 96, 1152), ("zh out", 7967, 96, 96),
          ("cross val x3", 178500, 96, 288), ("ffn1", 78899, 96, 192), ("ffn2", 78899, 192, 96)]


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_v = tot_o = 0.0
for name, T, K, N in shapes:
    x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d)
    y = torch.empty(T, N, device=d)
    us_v = timeit(lambda: torch.addmm(b, x, w.t()))
    us_o = timeit(lambda: linear_fwd(x, w, b, out=y))
    byt = 4 * (T * K + T * N + N * K)
    tot_v += us_v; tot_o += us_o
    print(f"{name:14s} T={T:6d} K={K:3d} N={N:4d}  vendor {us_v:7.1f} us  ours {us_o:7.1f} us  {byt / us_o / 1e3:7.1f} GB/s "
          f"{2 * T * K * N / us_o / 1e6:6.1f} TF/s  (bytes @4TB/s {byt / 4e6:5.1f} us, mfma @155TF {2 * T * K * N / 155e6:5.1f} us)")
print(f"sum vendor {tot_v:.0f} us  ours {tot_o:.0f} us")
# fused epilogues: output_proj + residual + LayerNorm vs addmm + add + layer_norm
T, K, N = 78899, 96, 96
x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d); r = torch.randn(T, N, device=d)
g = torch.ones(N, device=d); be = torch.zeros(N, device=d)
us_v = timeit(lambda: torch.nn.functional.layer_norm(torch.addmm(b, x, w.t()) + r, (N,), g, be))
us_o = timeit(lambda: linear_fwd(x, w, b, residual=r, ln=(g, be, 1e-5)))
print(f"proj + residual + LN (78899 x 96): torch {us_v:.1f} us  fused {us_o:.1f} us")
