"""GPU: every kernel family of the library beside a bf16-MFMA kernel on a second HIP stream.

Round 5 found that a compiler-formed, half-swapping packed-FP32 instruction returns wrong results while a wave issuing
``v_mfma_f32_16x16x32_bf16`` shares the SIMD; the regression test had only the MSDA forward kernels as victims.  This is the
matrix the round-5 review asked for: EVERY exported forward / backward entry point as the victim — render (fast, exact,
per-sample, backward with both scatters), field volume / query (both directions), the projections (forward with epilogues,
head-major, dgrad, wgrad), layer norm, reprojection + SSIM (both directions), eikonal / second differences, the Occ3D tail,
point sampling, MSDA (plain / fused / camera loop, both directions) — launched >= 100 times on stream B while
(1) ``selfocc_linear_fwd`` (bf16 x 3) and (2) ``selfocc_field_volume_bwd`` (bf16 x 3, 1 wave / SIMD, 160 KB LDS) loop on stream A.

POSITIVE CONTROL (added after round 6's first per-lease survey turned out to be blind): under the same schedule the isolated
half-swapping instruction form of scripts/micro/xlane_probe_lib.hip must come out WRONG beside ``selfocc_linear_fwd`` — proof
that victim and disturber waves really share SIMDs — while the safe forms stay right.

Criterion, self-calibrated per victim from three QUIET runs: bitwise equal to the quiet result when the quiet runs are bitwise
repeatable; otherwise (float atomics whose order the hardware picks: gradient scatters) within 8 x the quiet run-to-run
spread + 1e-6 of the tensor's scale.  A section-3.8 event is whole wrong rows of O(1) relative size — nowhere near either."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D0 = torch.device("cuda:0")
LAUNCHES = int(os.environ.get("SO_CONC_LAUNCHES", "100"))


def _flat(x):
    if torch.is_tensor(x):
        return [x]
    if isinstance(x, dict):
        return [t for k in sorted(x) for t in _flat(x[k])]
    if isinstance(x, (list, tuple)):
        return [t for y in x for t in _flat(y)]
    return []


def _victims():
    """name -> closure returning tensors; built once (module scope), inputs resident on the device"""
    from selfocc_amd import abi, synthetic as sy
    from selfocc_amd.render import render_rays, render_rays_autograd, SDFVolume, RaySet
    from selfocc_amd.field import field_volume, FieldVolumeFunction
    from selfocc_amd.occ import field_query, field_query_autograd, occ_resample, MeanIoU, OPENSEED2NUSCENES
    from selfocc_amd.linear import linear_fwd, linear_fwd_heads, linear_dgrad, linear_wgrad
    from selfocc_amd.reproj import ReprojSampleFunction
    from selfocc_amd.loss.reproj import SSIM
    from selfocc_amd.loss import EikonalLoss
    from selfocc_amd.msda import (msda_cross_inference, msda_fused_inference, multi_scale_deformable_attn, to_head_major,
                                  MSDAFusedFunction, MSDACrossFunction, MultiScaleDeformableAttnFunction)
    from selfocc_amd.model import bricks
    from selfocc_amd.model.encoder.utils import point_sampling
    g = torch.Generator(device=D0).manual_seed(17)
    rn = lambda *s: torch.randn(*s, device=D0, generator=g)
    ru = lambda *s: torch.rand(*s, device=D0, generator=g)
    V = {}
    # ---- render: the bench kernel (cfg2, C = 1), the exact path, C = 25, per-sample, backward with both scatters ----
    vol1 = sy.make_volume("cfg2", seed=3).to(D0)
    on_dev = lambda r: RaySet(img2lidar=r.img2lidar.to(D0), nx=r.nx, ny=r.ny, sx=r.sx, sy=r.sy, ox=r.ox, oy=r.oy)
    rays2 = on_dev(sy.make_rays("cfg2", seed=3))
    cfg2 = sy.make_render_config("cfg2", inv_s=20.0)
    V['render_fwd fast C=1 (bench frame)'] = lambda: render_rays(vol1, rays2, cfg2)
    vol25 = sy.make_volume("cfg5", n_rgb=3, n_sem=21, seed=4).to(D0)
    rays5 = on_dev(sy.make_rays("cfg5", seed=4))
    cfg5 = sy.make_render_config("cfg5", inv_s=20.0)
    cfg5x = sy.make_render_config("cfg5", inv_s=20.0, exact=True)
    V['render_fwd fast C=25'] = lambda: render_rays(vol25, rays5, cfg5)
    V['render_fwd exact C=25'] = lambda: render_rays(vol25, rays5, cfg5x)
    V['render_fwd per-sample (training API)'] = lambda: render_rays(vol25, rays5, cfg5, per_sample=True)
    inv_s = torch.tensor(20.0, device=D0)
    n5 = rays5.n_rays
    t_rand, bk = ru(n5), ru(n5, 3)
    G = dict(depth=rn(n5), acc=rn(n5), rgb=rn(n5, 3), sem=rn(n5, 21), weights=rn(n5, 256), sdf=0.1 * rn(n5, 256), grad=0.1 * rn(n5, 256, 3))

    def render_bwd(scatter):
        def run():
            c = sy.make_render_config("cfg5", inv_s=20.0, jitter_mode=abi.JITTER_SINGLE, bkgd_mode=abi.BKGD_PER_RAY)
            c.bwd_scatter = scatter
            sdf_p, feat_p, s_p = vol25.sdf.clone().requires_grad_(True), vol25.feat.clone().requires_grad_(True), inv_s.clone().requires_grad_(True)
            out = render_rays_autograd(SDFVolume(vol25.mapping, sdf_p, feat_p, 3, 21), s_p, rays5, c, t_rand=t_rand, bkgd_rays=bk)
            sum((out[k] * G[k]).sum() for k in G).backward()
            return sdf_p.grad, feat_p.grad, s_p.grad
        return run
    V['render_bwd atomic scatter'] = render_bwd('atomic')
    V['render_bwd binned scatter'] = render_bwd('binned')
    # ---- field volume / dense query ----
    H, W, Dd, C = 257, 257, 25, 96
    hw, zh, wz = rn(H * W, C), rn(Dd * H, C), rn(W * Dd, C)
    lins = [torch.nn.Linear(C, C).to(D0), torch.nn.Linear(C, 25).to(D0)]
    V['field_volume_fwd (bf16x3)'] = lambda: field_volume(hw, zh, wz, (H, W, Dd), lins, 24)
    g_sdf, g_feat = rn(H, W, Dd), rn(H, W, Dd, 24)

    def field_bwd():
        ps = [t.clone().requires_grad_(True) for t in (hw, zh, wz, lins[0].weight, lins[0].bias, lins[1].weight, lins[1].bias)]
        sdf, feat = FieldVolumeFunction.apply(*ps, (H, W, Dd), 24)
        torch.autograd.backward([sdf, feat], [g_sdf, g_feat])
        return [p.grad for p in ps]
    V['field_volume_bwd (bf16x3)'] = field_bwd
    xyz = (ru(200 * 200 * 16, 3) * torch.tensor([80.0, 80.0, 6.4], device=D0) + torch.tensor([-40.0, -40.0, -1.0], device=D0)).contiguous()
    V['field_query sdf + logits + argmax'] = lambda: field_query(vol25, xyz, want_sdf=True, want_logits=True, want_argmax=True)
    gq, gl = rn(xyz.shape[0]), rn(xyz.shape[0], 21)

    def query_bwd():
        sdf_p, feat_p = vol25.sdf.clone().requires_grad_(True), vol25.feat.clone().requires_grad_(True)
        q = field_query_autograd(SDFVolume(vol25.mapping, sdf_p, feat_p, 3, 21), xyz, want_logits=True)
        torch.autograd.backward([q['sdf'], q['logits']], [gq, gl])
        return sdf_p.grad, feat_p.grad
    V['field_query_bwd'] = query_bwd
    # ---- Occ3D tail ----
    grid = torch.stack(torch.meshgrid(torch.linspace(0.015, 0.985, 200), torch.linspace(0.02, 0.99, 200), torch.linspace(0.05, 0.95, 16),
                                      indexing='ij'), -1).to(D0).contiguous()
    sdf_d, logit_d = rn(200, 200, 16), rn(200, 200, 16, 21)
    gt = torch.randint(0, 18, (200, 200, 16), device=D0, generator=g).int()

    def occ_tail():
        got = occ_resample(sdf_d, grid, 0.0, logits=logit_d, lut=OPENSEED2NUSCENES, crop=(6, 6, 6, 6, 0, 4))
        m = MeanIoU(list(range(1, 17)), 0, [str(c) for c in range(1, 17)], False, 0)
        m.reset()
        m._after_step(got['sem'], gt)
        return got['occ'], got['sem'], m.counts
    V['occ_resample + iou_counts'] = occ_tail
    # ---- projections (the shipped layer's shapes), layer norm ----
    x = rn(78899, 96)
    w432, b432 = rn(432, 96) * 0.1, rn(432)
    w96, b96 = rn(96, 96) * 0.1, rn(96)
    res = rn(78899, 96)
    ln = torch.nn.LayerNorm(96).to(D0)
    V['linear_fwd bias'] = lambda: linear_fwd(x, w432, b432)
    V['linear_fwd bias + relu'] = lambda: linear_fwd(x, rn_w192, b192, relu=True)
    rn_w192, b192 = rn(192, 96) * 0.1, rn(192)
    V['linear_fwd residual + layernorm epilogue'] = lambda: linear_fwd(x, w96, b96, residual=res, ln=(ln.weight, ln.bias, ln.eps))
    xv = rn(6 * 25500, 96)
    V['linear_fwd_heads (head-major value)'] = lambda: linear_fwd_heads(xv, w96, b96, 25500)
    dy = rn(78899, 432)
    V['linear_dgrad'] = lambda: linear_dgrad(dy, w432)
    V['linear_wgrad'] = lambda: linear_wgrad(dy, x)
    fln = bricks.FastLayerNorm(96).to(D0)
    x3 = x[None]
    V['layernorm_fwd'] = lambda: fln(x3)
    g3 = rn(1, 78899, 96)

    def ln_bwd():
        xp = x3.clone().requires_grad_(True)
        fln.zero_grad(set_to_none=True)
        fln(xp).backward(g3)
        return xp.grad, fln.weight.grad, fln.bias.grad
    V['layernorm_bwd'] = ln_bwd
    # ---- losses: reprojection sampling, SSIM, eikonal, second differences ----
    R, S = 4800, 256
    wts, ts = torch.softmax(rn(R, S), -1), torch.cumsum(ru(R, S) * 0.3, -1) + 0.5
    pix = torch.stack([ru(R) * 1600, ru(R) * 768], -1)
    curr = ru(R, 3)
    K = torch.tensor([[1266.0, 0, 800, 0], [0, 1266.0, 384, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=D0)
    def motion(yaw, tx, tz):
        y = np.deg2rad(yaw)
        Rm = torch.tensor([[np.cos(y), 0, np.sin(y), tx], [0, 1, 0, 0], [-np.sin(y), 0, np.cos(y), tz], [0, 0, 0, 1]], device=D0, dtype=torch.float32)
        return K @ Rm @ torch.linalg.inv(K)
    Tp, Tn = motion(2, 0.3, -0.8), motion(-2, -0.3, 0.8)
    ip, inx = ru(3, 768, 1600), ru(3, 768, 1600)
    V['reproj_fwd'] = lambda: ReprojSampleFunction.apply(wts, ts, None, pix, curr, Tp, Tn, ip, inx, 768, 1600)
    gl1, gcomb = rn(R), rn(R, 3)

    def reproj_bwd():
        wp = wts.clone().requires_grad_(True)
        l1, comb, _ = ReprojSampleFunction.apply(wp, ts, None, pix, curr, Tp, Tn, ip, inx, 768, 1600)
        torch.autograd.backward([l1, comb], [gl1, gcomb])
        return wp.grad
    V['reproj_bwd'] = reproj_bwd
    sa, sb2 = ru(6, 3, 48, 100), ru(6, 3, 48, 100)
    ssim = SSIM()
    V['ssim_fwd'] = lambda: ssim(sa, sb2)
    gs = rn(6, 3, 48, 100)

    def ssim_bwd():
        a, b = sa.clone().requires_grad_(True), sb2.clone().requires_grad_(True)
        ssim(a, b).backward(gs)
        return a.grad, b.grad
    V['ssim_bwd'] = ssim_bwd
    eg = rn(6 * 4800 * 256, 3)
    eik = EikonalLoss(weight=0.1)

    def eik_both():
        e = eg.clone().requires_grad_(True)
        v = eik(dict(eik_grad=e))
        v.backward()
        return v.detach(), e.grad
    V['eikonal fwd + bwd'] = eik_both
    from selfocc_amd.model.head.neus_head import _SecondDiff

    def second_diff():
        s = vol25.sdf.clone().requires_grad_(True)
        o = _SecondDiff.apply(s)
        o.abs().mean().backward()
        return o.detach(), s.grad
    V['second_diff fwd + bwd'] = second_diff
    # ---- point sampling ----
    import hotpath_common as hc
    c2w, l2i, _K = hc.ring_cameras(6, (768, 1600), 1266.0)
    metas = [dict(lidar2img=l2i, img_shape=(768, 1600))]
    ref3d = (ru(1, 8, 66049, 3) * torch.tensor([80.0, 80.0, 6.4], device=D0) + torch.tensor([-40.0, -40.0, -1.0], device=D0)).contiguous()
    V['point_sampling'] = lambda: point_sampling(ref3d, metas)
    # ---- MSDA: plain / fused / camera loop, both directions (shipped hw-plane shapes) ----
    heads, d, cams, nq = 6, 16, 6, 66049
    shapes = torch.tensor([[96, 200], [48, 100], [24, 50], [12, 25]])
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    sh, st = shapes.to(D0), starts.to(D0)
    host = [int(v) for v in shapes.reshape(-1).tolist()]
    value = rn(cams, nv, heads, d)
    v_hm = to_head_major(value)
    L, P = 4, 8
    off, logits = rn(nq, heads, L, P, 2) * 2, rn(nq, heads, L * P)
    vis = ru(cams, nq) < 0.35
    refc, ref1 = ru(cams, nq, P, 2) * 1.2 - 0.1, ru(1, nq, P, 2) * 1.2 - 0.1
    nq2 = 22016
    loc = ru(1, nq2, heads, L, P, 2) * 1.1 - 0.05
    aw = torch.softmax(rn(1, nq2, heads, L * P), -1).view(1, nq2, heads, L, P)
    V['msda_cross_fwd'] = lambda: msda_cross_inference(v_hm, sh, st, refc, vis, off, logits, True)
    V['msda_fused_fwd'] = lambda: msda_fused_inference(v_hm[:1], sh, st, ref1, 1, off[None], logits[None], True)
    V['msda_fwd (mmcv op)'] = lambda: multi_scale_deformable_attn(value[:1], sh, st, loc, aw)
    go_q, go_2 = rn(nq, heads * d), rn(1, nq2, heads * d)

    def msda_bwd_plain():
        v, l, a = value[:1].clone().requires_grad_(True), loc.clone().requires_grad_(True), aw.clone().requires_grad_(True)
        MultiScaleDeformableAttnFunction.apply(v, sh, st, l, a, 64).backward(go_2)
        return v.grad, l.grad, a.grad
    V['msda_bwd (mmcv op)'] = msda_bwd_plain

    def msda_bwd_fused():
        v, o, lg = v_hm[:1].clone().requires_grad_(True), off[None].clone().requires_grad_(True), logits[None].clone().requires_grad_(True)
        MSDAFusedFunction.apply(v, sh, st, ref1, 1, o, lg, host, True, False).backward(go_q[None])
        return v.grad, o.grad, lg.grad
    V['msda_fused_bwd (point + band kernels)'] = msda_bwd_fused

    def msda_bwd_cross():
        v, o, lg = v_hm.clone().requires_grad_(True), off.clone().requires_grad_(True), logits.clone().requires_grad_(True)
        MSDACrossFunction.apply(v, sh, st, refc, vis, o, lg, host, True, False).backward(go_q)
        return v.grad, o.grad, lg.grad
    V['msda_cross_bwd (camera loop)'] = msda_bwd_cross
    # ---- the two disturbers ----
    dist = {'selfocc_linear_fwd (bf16x3)': lambda: linear_fwd(x, w432, b432),
            'selfocc_field_volume_bwd (bf16x3)': field_bwd}
    # ---- POSITIVE CONTROL: the instruction form that IS unsafe beside bf16 MFMA waves (`v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]`,
    # scripts/micro/xlane_probe_lib.hip: every result compared in-kernel with the unpacked op) must come out WRONG under the very
    # schedule that the 32 victims pass — otherwise a green matrix would only say that the harness never made kernels overlap
    # (round 6's first per-lease survey had exactly that flaw and reported 0 errors on 16 GPUs, scripts/pk_swizzle_survey.py) ----
    control = None
    so = os.path.join(ROOT, "scripts", "micro", "libxlane_probe.so")
    if not os.path.exists(so):          # built by __graft_entry__.build(); hipcc is on every box
        import subprocess
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pk_swizzle_survey.py"), "--build-only"])
    if os.path.exists(so):
        import ctypes as C
        probe = C.CDLL(so)
        table = torch.empty(1 << 22, 4, dtype=torch.int32, device=D0)
        probe.probe_fill_table(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(table.data_ptr()), 1 << 22)
        torch.cuda.synchronize()

        def control():
            cnt = torch.zeros(24, dtype=torch.int64, device=D0)
            probe.probe_victims(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(cnt.data_ptr()), 12345,
                                C.c_void_p(table.data_ptr()), 1 << 20)
            return cnt[:17]          # wrong results per instruction form (9, 11: the half-swapping ones); all zero when quiet
    return V, dist, control


_STATE = {}


def _setup():
    if not _STATE:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        _STATE['v'], _STATE['d'], _STATE['c'] = _victims()
    return _STATE['v'], _STATE['d'], _STATE['c']


def _run(fn):
    return [t.detach().clone() for t in _flat(fn()) if t is not None]


@pytest.mark.parametrize("disturber", ['selfocc_linear_fwd (bf16x3)', 'selfocc_field_volume_bwd (bf16x3)'])
def test_every_kernel_family_beside_a_bf16_mfma_kernel_on_another_stream(hip, disturber):
    victims, dists, control = _setup()
    dist = dists[disturber]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    report, failures = {}, {}
    for _ in range(3):
        dist()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dist()
    e1.record(); torch.cuda.synchronize()
    dist_ms = max(e0.elapsed_time(e1) / 10, 1e-3)
    for name, fn in victims.items():
        if name.startswith('field_volume_bwd') and disturber.startswith('selfocc_field_volume_bwd'):
            continue                                   # the disturber itself: covered beside the other one
        torch.cuda.synchronize()
        quiet = [_run(fn) for _ in range(3)]
        torch.cuda.synchronize()
        spread = [max(float((a.double() - b.double()).abs().max()) if a.numel() else 0.0 for a, b in ((q[i], quiet[0][i]) for q in quiet[1:]))
                  for i in range(len(quiet[0]))]
        exact = all(s == 0.0 for s in spread)
        # as many disturber launches per victim launch as it takes to keep stream A busy for the victim's whole duration
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); _run(fn); e1.record(); torch.cuda.synchronize()
        n_dist = min(64, int(e0.elapsed_time(e1) / dist_ms) + 2)
        bad, worst = 0, 0.0
        for it in range(LAUNCHES):
            with torch.cuda.stream(sa):
                for _ in range(n_dist):
                    dist()
            with torch.cuda.stream(sb):
                out = _run(fn)
            sb.synchronize()
            for i, (o, q) in enumerate(zip(out, quiet[0])):
                if exact:
                    ok = torch.equal(o, q)
                    dev = 0.0 if ok else float((o.double() - q.double()).abs().max())
                else:
                    scale = max(float(q.double().abs().max()), 1e-30) if q.numel() else 1.0
                    dev = float((o.double() - q.double()).abs().max()) if q.numel() else 0.0
                    ok = dev <= 8 * spread[i] + 1e-6 * scale and bool(torch.isfinite(o).all())
                worst = max(worst, dev)
                if not ok:
                    bad += 1
                    break
        torch.cuda.synchronize()
        report[name] = dict(mode='bitwise' if exact else 'atomic-order tolerance', quiet_spread=max(spread) if spread else 0.0,
                            bad_launches=bad, worst_dev=worst, disturber_launches_per_victim=n_dist)
        if bad:
            failures[name] = report[name]
    ctl = None
    if control is not None:
        torch.cuda.synchronize()
        quiet_ctl = control().clone()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); control(); e1.record(); torch.cuda.synchronize()
        n_dist = min(400, int(e0.elapsed_time(e1) / dist_ms * 1.25) + 4)
        wrong = torch.zeros(17, dtype=torch.int64, device=D0)
        rounds = max(4, LAUNCHES // 10)
        for it in range(rounds):
            with torch.cuda.stream(sa):
                for _ in range(n_dist):
                    dist()
            with torch.cuda.stream(sb):
                wrong += control()
            sb.synchronize()
        torch.cuda.synchronize()
        w = wrong.tolist()
        ctl = dict(rounds=rounds, disturber_launches_per_round=n_dist, quiet_wrong=int(quiet_ctl.sum()),
                   swizzled_wrong=w[9] + w[11], other_forms_wrong=sum(w) - w[9] - w[11], detected=bool(w[9] + w[11] > 0))
    try:
        import json
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "concurrency_matrix.jsonl"), "a") as f:
            f.write(json.dumps(dict(disturber=disturber, launches=LAUNCHES, positive_control=ctl, victims=report)) + "\n")
    except OSError:
        pass
    if ctl is not None:
        assert ctl['quiet_wrong'] == 0 and ctl['other_forms_wrong'] == 0, ctl       # the safe forms stay safe, the quiet run is clean
        # selfocc_linear_fwd's kernel shares SIMDs with other waves (<= 128 registers): the control MUST fail beside it (it does on
        # every device of the pool tried in round 6: 5 of 5, ~3 000 wrong results per round).  field_volume_bwd_b3 runs one wave per
        # SIMD with the whole register file (256 + 250 registers, 160 KB of LDS): no other wave can be resident on a SIMD it
        # occupies, and the control comes out clean beside it — measured, recorded, and the reason why that disturber says
        # nothing about this effect (it still exercises L2 / LDS / fabric contention for the 32 victims).
        if disturber.startswith('selfocc_linear_fwd') and not ctl['detected']:
            msg = (f"positive control NOT detected beside {disturber}: the half-swapping form came out right in {ctl['rounds']} rounds — "
                   "either this device does not show the effect or the schedule did not overlap the kernels; the green matrix is "
                   "inconclusive here (SO_CONC_REQUIRE_CONTROL=0 turns this into a warning)")
            if os.environ.get("SO_CONC_REQUIRE_CONTROL", "1") != "0":
                raise AssertionError(msg)
            import warnings
            warnings.warn(msg)
    assert not failures, f"victims that differ from their quiet result beside {disturber} (of {LAUNCHES} launches): {failures}"
    assert len(report) >= 30
