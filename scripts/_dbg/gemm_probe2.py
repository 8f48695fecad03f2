import torch
d = torch.device("cuda:0")
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
T = 257 * 257 * 25
x = torch.randn(T, 96, device=d); w1 = torch.randn(96, 96, device=d); w2 = torch.randn(25, 96, device=d); b2 = torch.randn(25, device=d)
dy2 = torch.randn(T, 25, device=d); dy1 = torch.randn(T, 96, device=d)
print("fwd addmm 96->25      ", t(lambda: torch.addmm(b2, x, w2.t())))
print("fwd x@w1^T 96->96      ", t(lambda: x @ w1.t()))
print("dx  dy2@w2 (T,25)@(25,96)", t(lambda: dy2 @ w2))
print("dx  dy1@w1 (T,96)@(96,96)", t(lambda: dy1 @ w1))
print("dW  dy2^T@x plain      ", t(lambda: dy2.t() @ x))
print("dW  dy1^T@x plain      ", t(lambda: dy1.t() @ x))
G = 256; Tp = (T // G) * G
print("dW  dy2 split-256      ", t(lambda: torch.bmm(dy2[:Tp].view(G, -1, 25).transpose(1, 2), x[:Tp].view(G, -1, 96)).sum(0)))
print("dW  dy1 split-256      ", t(lambda: torch.bmm(dy1[:Tp].view(G, -1, 96).transpose(1, 2), x[:Tp].view(G, -1, 96)).sum(0)))
print("db  dy2.sum(0)         ", t(lambda: dy2.sum(0)))
print("softplus               ", t(lambda: torch.nn.functional.softplus(x)))
