import torch, time
d = torch.device("cuda:0")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for T, i, o in [(78899, 96, 432), (78899, 96, 216), (78899, 96, 96), (78899, 96, 192), (78899, 192, 96),
                (153000, 96, 96), (6 * 22000, 96, 384), (6 * 22000, 96, 192), (6 * 4300, 96, 2304), (6 * 4300, 96, 1152)]:
    x = torch.randn(T, i, device=d); dy = torch.randn(T, o, device=d)
    plain = t(lambda: dy.t() @ x)
    G = 64
    Tp = (T // G) * G
    split = t(lambda: torch.bmm(dy[:Tp].view(G, -1, o).transpose(1, 2), x[:Tp].view(G, -1, i)).sum(0))
    alt = t(lambda: (x.t() @ dy).t())
    print(f"T={T:7d} in={i:4d} out={o:5d}  dY^T@X {plain:8.3f} ms   (X^T@dY)^T {alt:8.3f} ms   split-64 bmm {split:8.3f} ms")
