"""dev: dx = dy W at the training encoder's shapes: selfocc_linear_dgrad (SELFOCC_DGRAD_VARIANT) vs the vendor GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd.linear import linear_dgrad
d = torch.device("cuda:0"); torch.manual_seed(0)
tot = [0.0, 0.0]
for (T, N, K, cnt) in [(66049, 384, 96, 4), (66049, 192, 96, 4), (6425, 768, 96, 8), (6425, 384, 96, 8), (78899, 96, 96, 12), (78899, 48, 96, 4),
                       (78899, 192, 96, 4), (78899, 96, 192, 4), (153000, 96, 96, 4)]:
    dy = torch.randn(T, N, device=d); w = torch.randn(N, K, device=d)
    res = []
    for f in (lambda: linear_dgrad(dy, w), lambda: dy @ w):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    tot[0] += res[0] * cnt; tot[1] += res[1] * cnt
    print(f"V={os.environ.get('SELFOCC_DGRAD_VARIANT','0')} T={T} N={N} K={K}: ours {res[0]:.1f} us, vendor {res[1]:.1f} us")
print(f"weighted: ours {tot[0]/1e3:.2f} ms, vendor {tot[1]/1e3:.2f} ms")
