"""Dev: time linear_fwd on the layer's shapes with the library named by SELFOCC_HIP_LIB (A/B of diagnostic builds)."""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from selfocc_amd.linear import linear_fwd
d = torch.device("cuda:0")
out = {}
for name, T, K, N in [("self_ol", 78899, 96, 648), ("hw_ol", 66049, 96, 576), ("cross_val_x3", 178500, 96, 288), ("ffn1", 78899, 96, 192), ("self_val", 78899, 96, 96)]:
    x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d); y = torch.empty(T, N, device=d)
    for _ in range(5): linear_fwd(x, w, b, out=y)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): linear_fwd(x, w, b, out=y)
    e1.record(); torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / 50 * 1e3, 1)
print(os.path.basename(os.environ.get("SELFOCC_HIP_LIB", "shipped")), json.dumps(out))
# the FFN's second Linear + residual + LayerNorm (K = 192) and output_proj + residual + LayerNorm (K = 96)
for name, T, K, N in [("ffn2_ln", 78899, 192, 96), ("out_ln", 78899, 96, 96), ("ffn2", 78899, 192, 96)]:
    x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d) / K ** 0.5; b = torch.randn(N, device=d); r = torch.randn(T, N, device=d)
    g = torch.ones(N, device=d); be = torch.zeros(N, device=d)
    fn = (lambda: linear_fwd(x, w, b, residual=r, ln=(g, be, 1e-5))) if name.endswith("_ln") else (lambda: linear_fwd(x, w, b))
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) / 50 * 1e3, 1), "us", "SELFOCC_LINEAR_B3=" + os.environ.get("SELFOCC_LINEAR_B3", "1"))
