"""MSDA micro-benchmark at the reference's nuscenes_occ shapes: cross-attention of the TPV
planes (6 cameras, FPN 96x200 .. 12x25 = 25500 keys, 6 heads x 16, 4 levels, P = 8 / 48) and the cross-view
self-attention (78899 queries, 3 levels, P = 12).

Per kernel: HIP-event time and ALGORITHMIC GB/s — SURVEY §8(d): every distinct input byte once + every
API-visible output byte once — against the 8 TB/s HBM peak.
    plain     value + loc + attw + out                       (selfocc_msda_fwd / _bwd_banded)
    fused     value + ref + off_raw + logits + out           (selfocc_msda_fused_fwd / _fused_bwd)
    cross     value + ref + vis + off_raw + logits + out     (selfocc_msda_cross_fwd / _cross_bwd, camera loop)
    linears + kernel: the route the eval encoder takes (the two query Linears through selfocc_linear_fwd, then the kernel)
`--json`: one JSON object on the last line (bench.py's "roofline_msda").  `--case NAME`: one of the three shapes only
(scripts/pmc_msda_rows.sh profiles each shape in its own rocprofv3 session, so that a kernel's counters belong to one shape).
Every row also carries the texture-path view: `l1_gather_frac` (computed here) and — from profiles/pmc_msda.json, the
TA_BUSY_avr / TA_TA_BUSY_sum / GRBM_GUI_ACTIVE counters of the row's kernels in that shape — `ta_util` per kernel."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfocc_amd.msda import (MultiScaleDeformableAttnFunction as F, MSDAFusedFunction, MSDACrossFunction,
                              msda_fused_inference, msda_cross_inference, to_head_major)
from selfocc_amd.linear import linear_fwd

PEAK = 8000.0
d = torch.device("cuda:0")
torch.manual_seed(0)
FPN = [[96, 200], [48, 100], [24, 50], [12, 25]]
CASES = {
    # name: (bs, nq, shapes, P)
    "cross_hw": (6, 66049 // 3, FPN, 8),
    "cross_zh": (6, 6425 // 3 * 2, FPN, 48),
    "self_xview": (1, 78899, [[257, 257], [25, 257], [257, 25]], 12),
}


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def time_bwd(make_out, g, n=10):
    """backward only (forward outside the events); includes the zero-fill of grad_value."""
    tb = 0.0
    make_out().backward(g); torch.cuda.synchronize()      # warm-up: first-touch allocations of the gradients / workspace
    for _ in range(n):
        out = make_out()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out.backward(g); b.record(); torch.cuda.synchronize()
        tb += a.elapsed_time(b)
    return tb / n


L1_PEAK = 256 * 64 * 2.4          # GB/s: 256 CUs x 64 bytes / clk of vector L1 at 2.4 GHz


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONLY = sys.argv[sys.argv.index("--case") + 1] if "--case" in sys.argv else None
try:
    PMC = json.load(open(os.path.join(ROOT, "profiles", "pmc_msda.json")))
except OSError:
    PMC = {}
# the recorded counters are keyed by a hash of the kernels' sources (like profiles/pmc_traffic.json): counters taken on an
# older build of the kernels are not attached to rows timed on this one
sys.path.insert(0, ROOT)
import bench as _bench
PMC_STALE = bool(PMC) and PMC.get("sources_sha1") != _bench.sources_hash(_bench.MSDA_SOURCES)
if PMC_STALE:
    PMC = {"source": f"{PMC.get('source')}: measured on kernel sources {PMC.get('sources_sha1')} != current "
                     f"{_bench.sources_hash(_bench.MSDA_SOURCES)} — not attached", "cases": {}}
# row label -> the HIP kernels it launches (function names; the gathers are in the first one)
HIP_KERNELS = (("linears + msda_cross_fwd", ["msda_cross_fwd_kernel"]),
               ("linears + msda_fused_fwd", ["msda_fused_fwd_kernel"]),
               ("msda_cross_bwd", ["msda_cross_bwd_point_kernel", "msda_bwd_band_list_kernel"]), ("msda_cross_fwd", ["msda_cross_fwd_kernel"]),
               ("msda_fused_bwd", ["msda_fused_bwd_point_kernel", "msda_bwd_band_list_kernel"]), ("msda_fused_fwd", ["msda_fused_fwd_kernel"]),
               ("msda_bwd", ["msda_bwd_point_kernel", "msda_bwd_band_list_kernel"]), ("msda_fwd", ["msda_fwd_kernel"]))


def ta_view(kernel, case):
    """TA utilisation of the row's kernels in this shape, from the committed PMC record (None when not recorded):
    ta_util = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8 XCDs) — the average texture addresser's busy share of the kernel;
    ta_unit_util = TA_TA_BUSY_sum / 256 TAs over the same cycles."""
    names = next((v for k, v in HIP_KERNELS if kernel.startswith(k)), [])
    out = {}
    for n in names:
        for full, c in PMC.get("cases", {}).get(case, {}).items():
            if full.startswith(n) and c.get("gui_active"):
                per_xcd = c["gui_active"] / 8.0
                out[full] = dict(ta_util=round(c["ta_busy_avr"] / per_xcd, 3), ta_unit_util=round(c["ta_ta_busy_sum"] / 256.0 / per_xcd, 3),
                                 launches=c.get("n"))
    return out or None


def rec(kernel, shape, alg_bytes, ms, points):
    gbps = alg_bytes / ms / 1e6
    # the texture path: a bilinear sample is 4 corners x 64 bytes of float32 channels through the CU's vector L1, whatever the
    # caches behind it do (an upper bound on the gathered bytes: samples outside the map and invisible pairs gather nothing)
    l1 = points * 256.0 / ms / 1e6
    case = "cross_hw" if "camera loop" in shape else shape.split(":")[0]
    return dict(kernel=kernel, shape=shape, alg_MB=round(alg_bytes / 1e6, 1), ms=round(ms, 4), GBps=round(gbps, 1),
                frac_of_8TBps=round(gbps / PEAK, 4), Gpoints_per_s=round(points / ms / 1e6, 2),
                l1_gather_GBps=round(l1, 0), l1_gather_frac=round(l1 / L1_PEAK, 3), ta=ta_view(kernel, case))


def query_linears(nq, H, L, P):
    """query features ~ N(0, 1) and the two Linears, scaled so that offsets ~ N(0, 2 px) and logits ~ N(0, 1)"""
    x = torch.randn(nq, 96, device=d)
    lo = torch.nn.Linear(96, H * L * P * 2).to(d)
    la = torch.nn.Linear(96, H * L * P).to(d)
    with torch.no_grad():
        lo.weight.normal_(0, 2.0 / 96 ** 0.5); lo.bias.zero_()
        la.weight.normal_(0, 1.0 / 96 ** 0.5); la.bias.zero_()
    return x, lo, la


rows = []
for name, (bs, nq, shapes, P) in CASES.items():
    if ONLY and name != ONLY:
        continue
    sh = torch.tensor(shapes, device=d)
    sh._so_host = [int(v) for v in sh.reshape(-1).tolist()]
    st = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
    nv = int((sh[:, 0] * sh[:, 1]).sum()); L = len(shapes); H = 6; D = 16
    value = torch.randn(bs, nv, H, D, device=d)
    # projected-like locality: neighbouring queries -> neighbouring pixels, + N(0, 2px) offsets
    side = int(nq ** 0.5) + 1
    qi = torch.arange(nq, device=d)
    base = torch.stack([(qi % side) / side, (qi // side) / side], -1)            # nq, 2
    off_raw = torch.randn(bs, nq, H, L, P, 2, device=d) * 2.0                     # pixels
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).float()                            # (L, 2) = (W, H)
    loc = base[None, :, None, None, None, :] + off_raw / wh[None, None, None, :, None, :]
    logits = torch.randn(bs, nq, H, L * P, device=d)
    attw = torch.softmax(logits, -1).view(bs, nq, H, L, P)
    pts = bs * nq * H * L * P
    tag = f"{name}: bs={bs} nq={nq} L={L} P={P} heads=6x16 keys={nv}"
    # ---- plain op (the mmcv boundary) ----
    value.requires_grad_(True); loc.requires_grad_(True); attw.requires_grad_(True)
    out = F.apply(value, sh, st, loc, attw, 64)
    g = torch.randn_like(out)
    with torch.no_grad():
        fwd_ms = timeit(lambda: F.apply(value, sh, st, loc, attw, 64))
    bwd_ms = time_bwd(lambda: F.apply(value, sh, st, loc, attw, 64), g)
    alg_f = 4 * (value.numel() + loc.numel() + attw.numel() + out.numel())
    alg_b = 4 * (value.numel() * 2 + loc.numel() * 2 + attw.numel() * 2 + out.numel())
    rows.append(rec("msda_fwd (plain)", tag, alg_f, fwd_ms, pts))
    rows.append(rec("msda_bwd point+band (plain, incl. grad_value memset)", tag, alg_b, bwd_ms, pts))
    # ---- fused prologue (softmax + loc inside the kernel) ----
    ref = base[None, :, None, :].expand(bs, nq, L, 2).contiguous()               # ref_kind 0: (bs, nq, L, 2)
    off_d = off_raw.detach().clone().requires_grad_(True)
    lg_d = logits.detach().clone().requires_grad_(True)
    val_d = value.detach().clone().requires_grad_(True)
    try:
        with torch.no_grad():
            f_ms = timeit(lambda: msda_fused_inference(val_d, sh, st, ref, 0, off_d, lg_d))
        alg_ff = 4 * (value.numel() + ref.numel() + off_raw.numel() + logits.numel() + out.numel())
        rows.append(rec("msda_fused_fwd", tag, alg_ff, f_ms, pts))
        fb_ms = time_bwd(lambda: MSDAFusedFunction.apply(val_d, sh, st, ref, 0, off_d, lg_d, sh._so_host), g)
        alg_fb = 4 * (value.numel() * 2 + ref.numel() + off_raw.numel() * 2 + logits.numel() * 2 + out.numel())
        rows.append(rec("msda_fused_bwd point+band (incl. grad_value memset)", tag, alg_fb, fb_ms, pts))
        # the layout the encoder modules use: head-major value (bs, heads, nv, d), same bytes
        val_h = to_head_major(val_d.detach()).requires_grad_(True)
        with torch.no_grad():
            fh_ms = timeit(lambda: msda_fused_inference(val_h, sh, st, ref, 0, off_d, lg_d, True))
        rows.append(rec("msda_fused_fwd head-major", tag, alg_ff, fh_ms, pts))
        fhb_ms = time_bwd(lambda: MSDAFusedFunction.apply(val_h, sh, st, ref, 0, off_d, lg_d, sh._so_host, True), g)
        rows.append(rec("msda_fused_bwd head-major point+band (incl. grad_value memset)", tag, alg_fb, fhb_ms, pts))
        if bs == 1 and 80 < 3 * L * P <= 112:
            # ---- the eval encoder's route: the two query linears + the fused kernel ----
            xq, lo, la = query_linears(nq, H, L, P)
            with torch.no_grad():
                def separate():
                    o = linear_fwd(xq, lo.weight, lo.bias).view(1, nq, H, L, P, 2)
                    lgt = linear_fwd(xq, la.weight, la.bias).view(1, nq, H, L * P)
                    return msda_fused_inference(val_h, sh, st, ref, 0, o, lgt, True)
                sep_ms = timeit(separate)
            wbytes = 4 * (lo.weight.numel() + la.weight.numel() + lo.bias.numel() + la.bias.numel())
            alg_sep = alg_ff + 4 * xq.numel() + wbytes + 4 * (off_raw.numel() + logits.numel())   # off / logits written once more
            rows.append(rec("linears + msda_fused_fwd head-major (separate route)", tag, alg_sep, sep_ms, pts))
    except Exception as e:   # a shape the fused / banded path does not take: keep the other rows
        rows.append(dict(kernel="msda_fused", shape=tag, error=repr(e)[:200]))
    if name == "cross_hw":
        # ---- camera loop: BEVCrossAttention without the re-batch; every query visible in ~2 of 6 cameras ----
        nq_full = 66049
        # spatially coherent visibility, as point_sampling produces: camera c sees a contiguous angular sector of
        # the BEV plane, neighbouring sectors overlap (every query is seen by 2 of the 6 cameras)
        side_f = 257
        qy, qx = torch.div(torch.arange(nq_full, device=d), side_f, rounding_mode='floor'), torch.arange(nq_full, device=d) % side_f
        ang = torch.atan2(qy.float() - 128, qx.float() - 128)                        # (-pi, pi]
        sector = ((ang + 3.14159265) / (2 * 3.14159265) * 6).clamp(0, 5.999)
        cam_ids = torch.arange(bs, device=d)[:, None].float()
        dist_c = torch.remainder(sector[None] - cam_ids - 0.5, 6.0)
        vis = (dist_c < 1.0) | (dist_c > 5.0)
        refc = torch.rand(bs, nq_full, P, 2, device=d)
        offc = (torch.randn(nq_full, H, L, P, 2, device=d) * 2.0).requires_grad_(True)
        lgc = torch.randn(nq_full, H, L * P, device=d).requires_grad_(True)
        with torch.no_grad():
            c_ms = timeit(lambda: msda_cross_inference(val_d, sh, st, refc, vis, offc, lgc))
        outc = nq_full * H * D
        alg_c = 4 * (value.numel() + refc.numel() + offc.numel() + lgc.numel() + outc) + vis.numel()
        ptsc = int(vis.sum().item()) * H * L * P
        tagc = f"cross_hw camera loop: cams=6 nq={nq_full} (visible pairs {int(vis.sum().item())}) L=4 P=8"
        rows.append(rec("msda_cross_fwd", tagc, alg_c, c_ms, ptsc))
        gc = torch.randn(nq_full, H * D, device=d)
        cb_ms = time_bwd(lambda: MSDACrossFunction.apply(val_d, sh, st, refc, vis, offc, lgc, sh._so_host), gc)
        alg_cb = 4 * (value.numel() * 2 + refc.numel() + offc.numel() * 2 + lgc.numel() * 2 + outc) + vis.numel()
        rows.append(rec("msda_cross_bwd point+band (incl. grad_value memset)", tagc, alg_cb, cb_ms, ptsc))
        with torch.no_grad():
            ch_ms = timeit(lambda: msda_cross_inference(val_h, sh, st, refc, vis, offc, lgc, True))
        rows.append(rec("msda_cross_fwd head-major", tagc, alg_c, ch_ms, ptsc))
        chb_ms = time_bwd(lambda: MSDACrossFunction.apply(val_h, sh, st, refc, vis, offc, lgc, sh._so_host, True), gc)
        rows.append(rec("msda_cross_bwd head-major point+band (incl. grad_value memset)", tagc, alg_cb, chb_ms, ptsc))
        if True:
            xq, lo, la = query_linears(nq_full, H, L, P)
            with torch.no_grad():
                def separate_c():
                    o = linear_fwd(xq, lo.weight, lo.bias).view(nq_full, H, L, P, 2)
                    lgt = linear_fwd(xq, la.weight, la.bias).view(nq_full, H, L * P)
                    return msda_cross_inference(val_h, sh, st, refc, vis, o, lgt, True)
                sepc_ms = timeit(separate_c)
            wbytes = 4 * (lo.weight.numel() + la.weight.numel() + lo.bias.numel() + la.bias.numel())
            alg_sepc = alg_c + 4 * xq.numel() + wbytes + 4 * (offc.numel() + lgc.numel())
            rows.append(rec("linears + msda_cross_fwd head-major (separate route)", tagc, alg_sepc, sepc_ms, ptsc))
    for r in rows[-16:]:
        if name in r["shape"] or "camera loop" in r["shape"]:
            print(json.dumps(r), flush=True)

print(json.dumps({"peak_GBps": PEAK, "bound": "hbm",
                  "definition": "algorithmic bytes (SURVEY 8d: inputs once + outputs once) / HIP-event time per call",
                  "l1_definition": "second view, the bound these gather kernels actually sit under: points x 4 corners x 64 B through the "
                                   "vector L1 (64 B / clk / CU x 256 CUs x 2.4 GHz = 39.3 TB/s) / the same time; forward rows only read "
                                   "the corners once, backward rows read them once and scatter them once",
                  "ta_definition": "per row, per HIP kernel: ta_util = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8 XCDs), ta_unit_util = TA_TA_BUSY_sum / "
                                   "256 / the same cycles; RECORDED counters (" + str(PMC.get("source")) + "), not taken in this run",
                  "kernels": rows}), flush=True)
