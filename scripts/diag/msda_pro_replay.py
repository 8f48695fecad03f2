"""DESIGN.md section 3.8, the loose end of round 5: the prologue-fused MSDA forward (`selfocc_msda_pro_fwd`, removed with ABI 31)
in its PLAIN form — the round-5 source (git 0d60026^: csrc/msda_pro.hip + its headers) compiled with the FINAL toolchain flags
(-fno-slp-vectorize -fno-vectorize: 0 v_pk_* instructions, checked with llvm-objdump) into scripts/diag/libmsda_pro_diag.so —
"still showed one wrong row in some launches" and was deleted rather than explained.  This replays it: N launches of the two
shipped shapes it served (hw-plane camera loop, cross-view self-attention), every result compared BITWISE with the first and
within float32 rounding with the separate route (selfocc_linear_fwd x 2 + selfocc_msda_fused / _cross_fwd of the current library).
The kernel has its bf16-MFMA prologue and its gather stage in ONE block (producer and consumer waves share SIMDs), so no second
stream is needed.  Build (where the git history is):
    d=$(mktemp -d); for f in msda_pro.hip msda_device.h so_device.h common.hip; do git show 0d60026^:selfocc_amd/csrc/$f > $d/$f; done
    (the headers' relative include of include/selfocc_hip.h: two directories up) ; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC
    -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-vectorize -shared msda_pro.hip common.hip -o scripts/diag/libmsda_pro_diag.so
    (libmsda_pro_diag_packed.so: the same without the two -fno-* flags = the round-5 build, 2 469 v_pk_* of which 128 half-swapping
    v_pk_mul_f32: `hipcc -S` + grep; the replay's positive control)
usage: python scripts/diag/msda_pro_replay.py [launches per shape, default 10000] [plain | packed]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from selfocc_amd.linear import linear_fwd
from selfocc_amd.msda import msda_fused_inference, msda_cross_inference, to_head_major

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
VARIANT = sys.argv[2] if len(sys.argv) > 2 else "plain"      # "packed": the same source with the compiler's vectorizers ON (the round-5 build: 128 half-swapping v_pk_mul_f32) — the replay's positive control
d0 = torch.device("cuda:0")
so = C.CDLL(os.path.join(ROOT, "scripts", "diag", "libmsda_pro_diag.so" if VARIANT == "plain" else "libmsda_pro_diag_packed.so"))
so.selfocc_last_error.restype = C.c_char_p
g = torch.Generator(device=d0).manual_seed(5)
rn = lambda *s: torch.randn(*s, device=d0, generator=g)
ru = lambda *s: torch.rand(*s, device=d0, generator=g)
P_ = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
heads, d, K = 6, 16, 96
res = {}


def pro(value_hm, sh, st, ref, ref_kind, vis, x, w_off, b_off, w_aw, b_aw, out, cams, bs, nv, nq, L, Pp):
    rc = so.selfocc_msda_pro_fwd(P_(value_hm), P_(sh), P_(st), P_(ref), ref_kind, P_(vis), P_(x), P_(w_off), P_(b_off), P_(w_aw), P_(b_aw),
                                 P_(out), cams, bs, nv, nq, heads, d, L, Pp, K, 0, 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, so.selfocc_last_error()


with torch.no_grad():
    for name in ("hw-plane camera loop", "cross-view self-attention"):
        if name.startswith("hw"):
            cams, nq, L, Pp = 6, 66049, 4, 8
            shapes = torch.tensor([[96, 200], [48, 100], [24, 50], [12, 25]])
        else:
            cams, nq, L, Pp = 0, 78899, 3, 12
            shapes = torch.tensor([[257, 257], [25, 257], [257, 25]])
        starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
        nv = int((shapes[:, 0] * shapes[:, 1]).sum())
        sh, st = shapes.to(d0).int(), starts.to(d0).int()
        nb = max(cams, 1)
        value_hm = to_head_major(rn(nb, nv, heads, d)).contiguous()
        x = rn(nq, K)
        w_off, b_off = rn(heads * L * Pp * 2, K) * 0.05, rn(heads * L * Pp * 2) * 2
        w_aw, b_aw = rn(heads * L * Pp, K) * 0.1, rn(heads * L * Pp)
        if cams:
            ref = (ru(cams, nq, Pp, 2) * 1.2 - 0.1).contiguous()
            vis = (ru(cams, nq) < 0.35).to(torch.uint8).contiguous()
            kind = 1
        else:
            ref = (ru(1, nq, L, Pp, 2) * 1.1 - 0.05).contiguous()
            vis, kind = None, 2
        out = torch.empty(nq, heads * d, device=d0)
        pro(value_hm, sh, st, ref, kind, vis, x, w_off, b_off, w_aw, b_aw, out, cams, 1, nv, nq, L, Pp)
        torch.cuda.synchronize()
        first = out.clone()
        # the separate route of the current library
        off = linear_fwd(x, w_off, b_off).view(nq, heads, L, Pp, 2)
        lg = linear_fwd(x, w_aw, b_aw).view(nq, heads, L * Pp)
        if cams:
            sep = msda_cross_inference(value_hm, sh, st, ref, vis.bool(), off, lg, True)
        else:
            sep = msda_fused_inference(value_hm, sh, st, ref, kind, off[None], lg[None], True)[0]
        torch.cuda.synchronize()
        scale = float(sep.abs().max())
        vs_sep = float((first - sep).abs().max()) / scale
        bad_launches, bad_rows, examples = 0, 0, []
        t0 = time.time()
        chunk = 50
        for it in range(0, N, chunk):
            outs = []
            for _ in range(chunk):
                o = torch.empty_like(first)
                pro(value_hm, sh, st, ref, kind, vis, x, w_off, b_off, w_aw, b_aw, o, cams, 1, nv, nq, L, Pp)
                outs.append(o)
            torch.cuda.synchronize()
            for j, o in enumerate(outs):
                if not torch.equal(o, first):
                    rows = (o.view(nq * heads, d) != first.view(nq * heads, d)).any(-1).nonzero().flatten()
                    bad_launches += 1
                    bad_rows += int(rows.numel())
                    if len(examples) < 5:
                        r = int(rows[0])
                        examples.append(dict(launch=it + j, n_rows=int(rows.numel()), row=r, query=r // heads, head=r % heads,
                                             got=[round(float(v), 5) for v in o.view(-1, d)[r][:6]],
                                             want=[round(float(v), 5) for v in first.view(-1, d)[r][:6]]))
        res[name] = dict(launches=N, launches_differing_from_the_first=bad_launches, wrong_rows_total=bad_rows,
                         max_rel_diff_vs_separate_route=vs_sep, seconds=round(time.time() - t0, 1), examples=examples)
p = torch.cuda.get_device_properties(0)
res['variant'] = VARIANT
res['device'] = dict(uuid=str(getattr(p, 'uuid', '')), pci=f"{getattr(p, 'pci_bus_id', 0):02x}", name=p.name)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "msda_pro_replay.jsonl"), "a") as f:
    f.write(json.dumps(res) + "\n")
print(json.dumps(res, indent=1))
