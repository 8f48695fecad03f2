"""selfocc_linear_fwd vs torch.addmm (hipBLASLt) on the encoder's Linear shapes (nuscenes_depth, dim 96): HIP-event
times, float32-MFMA rate against the 157.3 TFLOP/s peak (MI355X_MICROARCH.md) and bytes moved; --json prints one JSON
line (bench.py's "roofline_linear")."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfocc_amd.linear import linear_fwd

d = torch.device("cuda:0")
PEAK_TF = 157.3
shapes = [("self_off", 78899, 96, 432), ("self_aw", 78899, 96, 216), ("self_val", 78899, 96, 96),
          ("hw_off", 66049, 96, 384), ("hw_aw", 66049, 96, 192), ("hw_out", 66049, 96, 96),
          ("zh_off", 7967, 96, 2304), ("zh_aw", 7967, 96, 1152), ("zh_out", 7967, 96, 96),
          ("cross_val_x3", 178500, 96, 288), ("ffn1", 78899, 96, 192), ("ffn2", 78899, 192, 96)]


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


as_json = "--json" in sys.argv
res = {"bound": "mfma", "peak": PEAK_TF, "unit": "TFLOP/s", "dtype": "f32 (v_mfma_f32_16x16x4_f32)", "shapes": {}}
tot_v = tot_o = flops = 0.0
for name, T, K, N in shapes:
    x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d)
    y = torch.empty(T, N, device=d)
    us_v = timeit(lambda: torch.addmm(b, x, w.t()))
    us_o = timeit(lambda: linear_fwd(x, w, b, out=y))
    byt = 4 * (T * K + T * N + N * K)
    tot_v += us_v; tot_o += us_o; flops += 2.0 * T * K * N
    tf = 2 * T * K * N / us_o / 1e6
    res["shapes"][name] = {"T": T, "K": K, "N": N, "us": round(us_o, 1), "vendor_us": round(us_v, 1), "achieved": round(tf, 1),
                           "frac": round(tf / PEAK_TF, 3), "alg_GBps": round(byt / us_o / 1e3, 0)}
    if not as_json:
        print(f"{name:14s} T={T:6d} K={K:3d} N={N:4d}  vendor {us_v:7.1f} us  ours {us_o:7.1f} us  {byt / us_o / 1e3:7.1f} GB/s "
              f"{tf:6.1f} TF/s  (bytes @4TB/s {byt / 4e6:5.1f} us, mfma @155TF {2 * T * K * N / 155e6:5.1f} us)")
# fused epilogues: output_proj + residual + LayerNorm vs addmm + add + layer_norm
T, K, N = 78899, 96, 96
x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d); r = torch.randn(T, N, device=d)
g = torch.ones(N, device=d); be = torch.zeros(N, device=d)
us_v = timeit(lambda: torch.nn.functional.layer_norm(torch.addmm(b, x, w.t()) + r, (N,), g, be))
us_o = timeit(lambda: linear_fwd(x, w, b, residual=r, ln=(g, be, 1e-5)))
res["layer_sum"] = {"us": round(tot_o, 1), "vendor_us": round(tot_v, 1), "achieved": round(flops / tot_o / 1e6, 1),
                    "frac": round(flops / tot_o / 1e6 / PEAK_TF, 3)}
res["proj_residual_layernorm_78899x96"] = {"us": round(us_o, 1), "torch_3_kernels_us": round(us_v, 1),
                                           "alg_GBps": round(4 * (3 * T * N) / us_o / 1e3, 0)}
if as_json:
    print(json.dumps(res))
else:
    print(f"sum vendor {tot_v:.0f} us  ours {tot_o:.0f} us  ({flops / tot_o / 1e6:.1f} TF/s)")
    print(f"proj + residual + LN (78899 x 96): torch {us_v:.1f} us  fused {us_o:.1f} us")
