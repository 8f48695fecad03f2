"""The encoder's projections (selfocc_linear_fwd: `linear_fwd_b3_kernel`, float32 through an exact three-way bfloat16 split on
the bf16 matrix pipe) on the shapes ONE shipped TPVFormerLayer runs in inference (nuscenes_depth, dim 96), against torch.addmm
(hipBLASLt).  These GEMMs have K = 96: 2 * 96 flops per 4-byte output — they are HBM-bound, so the roofline is bytes:
    bound "hbm", achieved = (x + W + y bytes) / HIP-event time, peak 8 TB/s (MI355X_MICROARCH.md), frac = achieved / peak;
the matrix-pipe view is secondary: flops x 3 (bf16 products per float32 product, of the 6 issued) against the dense bf16 peak
2.5 PFLOP/s, i.e. a float32-equivalent ceiling of 2500 / 6 = 417 TFLOP/s (round 5 printed `frac` against the 157 TFLOP/s of
v_mfma_f32_16x16x4_f32 — the kernel that runs has not used that instruction since round 3).
Round 6: `sampling_offsets` | `attention_weights` are ONE projection per attention (bricks.merged_off_logits): the rows
`self_ol`, `hw_ol`, `zh_ol` replace the pairs `*_off` + `*_aw` (also timed, for the A/B in `merged_vs_pairs_us`).
--json prints one JSON line (bench.py's "roofline_linear")."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfocc_amd.linear import linear_fwd

d = torch.device("cuda:0")
HBM_PEAK = 8000.0            # GB/s
MATRIX_CEIL_TF = 2500.0 / 6  # float32-equivalent TFLOP/s of the bf16 x 3 scheme (six bf16 MFMAs per float32 k-step)
layer = [("self_ol", 78899, 96, 648), ("self_val", 78899, 96, 96), ("hw_ol", 66049, 96, 576), ("hw_out", 66049, 96, 96),
         ("zh_ol", 7967, 96, 3456), ("zh_out", 7967, 96, 96), ("cross_val_x3", 178500, 96, 288), ("ffn1", 78899, 96, 192),
         ("ffn2", 78899, 192, 96)]
pairs = [("self_off", 78899, 96, 432), ("self_aw", 78899, 96, 216), ("hw_off", 66049, 96, 384), ("hw_aw", 66049, 96, 192),
         ("zh_off", 7967, 96, 2304), ("zh_aw", 7967, 96, 1152)]


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(T, K, N):
    x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d)
    y = torch.empty(T, N, device=d)
    us_v = timeit(lambda: torch.addmm(b, x, w.t()))
    us_o = timeit(lambda: linear_fwd(x, w, b, out=y))
    return us_o, us_v, 4 * (T * K + T * N + N * K), 2.0 * T * K * N


as_json = "--json" in sys.argv
res = {"bound": "hbm", "peak": HBM_PEAK, "unit": "GB/s", "kernel": "linear_fwd_b3_kernel (float32 = 3 x bfloat16, six v_mfma_f32_16x16x32_bf16 per k-step)",
       "matrix_ceiling_TFLOPs_f32_equiv": round(MATRIX_CEIL_TF, 1), "shapes": {}}
tot_v = tot_o = flops = byts = 0.0
for name, T, K, N in layer:
    us_o, us_v, byt, fl = run(T, K, N)
    tot_v += us_v; tot_o += us_o; flops += fl; byts += byt
    gbps = byt / us_o / 1e3
    res["shapes"][name] = {"T": T, "K": K, "N": N, "us": round(us_o, 1), "vendor_us": round(us_v, 1), "achieved": round(gbps, 0),
                           "frac": round(gbps / HBM_PEAK, 3), "matrix_frac": round(fl / us_o / 1e6 / MATRIX_CEIL_TF, 3)}
    if not as_json:
        print(f"{name:14s} T={T:6d} K={K:3d} N={N:4d}  vendor {us_v:7.1f} us  ours {us_o:7.1f} us  {gbps:7.0f} GB/s = {gbps / HBM_PEAK:.2f} of HBM   "
              f"{fl / us_o / 1e6:6.1f} TF/s = {fl / us_o / 1e6 / MATRIX_CEIL_TF:.2f} of the bf16x3 ceiling")
pair_us = {}
for name, T, K, N in pairs:
    pair_us[name] = run(T, K, N)[0]
res["merged_vs_pairs_us"] = {p: {"merged": res["shapes"][p + "_ol"]["us"], "pair": round(pair_us[p + "_off"] + pair_us[p + "_aw"], 1)}
                             for p in ("self", "hw", "zh")}
# fused epilogues: output_proj + residual + LayerNorm vs addmm + add + layer_norm
T, K, N = 78899, 96, 96
x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d); r = torch.randn(T, N, device=d)
g = torch.ones(N, device=d); be = torch.zeros(N, device=d)
us_v = timeit(lambda: torch.nn.functional.layer_norm(torch.addmm(b, x, w.t()) + r, (N,), g, be))
us_o = timeit(lambda: linear_fwd(x, w, b, residual=r, ln=(g, be, 1e-5)))
res["layer_sum"] = {"us": round(tot_o, 1), "vendor_us": round(tot_v, 1), "achieved": round(byts / tot_o / 1e3, 0),
                    "frac": round(byts / tot_o / 1e3 / HBM_PEAK, 3), "matrix_frac": round(flops / tot_o / 1e6 / MATRIX_CEIL_TF, 3),
                    "pairs_instead_of_merged_us": round(tot_o - sum(res["shapes"][p + "_ol"]["us"] for p in ("self", "hw", "zh")) + sum(pair_us.values()), 1)}
res["proj_residual_layernorm_78899x96"] = {"us": round(us_o, 1), "torch_3_kernels_us": round(us_v, 1),
                                           "achieved": round(4 * (3 * T * N) / us_o / 1e3, 0)}
if as_json:
    print(json.dumps(res))
else:
    print(f"layer sum: vendor {tot_v:.0f} us  ours {tot_o:.0f} us  = {byts / tot_o / 1e3:.0f} GB/s ({byts / tot_o / 1e3 / HBM_PEAK:.2f} of HBM); "
          f"with the off / aw pairs instead of the merged rows {res['layer_sum']['pairs_instead_of_merged_us']:.0f} us")
    print(f"proj + residual + LN (78899 x 96): torch {us_v:.1f} us  fused {us_o:.1f} us")
