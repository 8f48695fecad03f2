"""GPU parity: selfocc_msda_fwd/_bwd (HIP, through the C ABI + autograd Function) vs the C oracle."""
import pytest
import torch

import oracle
from selfocc_amd.msda import MultiScaleDeformableAttnFunction
from test_oracle_msda_cpu import make_case, CASES

pytestmark = pytest.mark.gpu

GPU_CASES = CASES + [
    (1, 300, 6, 16, [[24, 50], [12, 25], [6, 13], [3, 7]], 48),   # P = 48 (zh / wz planes): 192 points / head
    (6, 200, 6, 16, [[24, 50], [12, 25], [6, 13], [3, 7]], 8),    # 6 cameras as batch
    (1, 1, 1, 4, [[2, 2]], 1),                                      # smallest
    (1, 3, 3, 16, [[3, 3]], 70),                                    # LP > 64
]


@pytest.mark.parametrize("mode", ["banded", "atomic"])
@pytest.mark.parametrize("case", GPU_CASES)
def test_msda_fwd_bwd_vs_oracle(hip, case, mode, monkeypatch):
    import selfocc_amd.msda as M
    monkeypatch.setattr(M, "BACKWARD_MODE", mode)
    value, shapes, starts, loc, attw = make_case(*case, seed=5)
    d = torch.device("cuda:0")
    v = value.to(d).requires_grad_(True); lc = loc.to(d).requires_grad_(True); aw = attw.to(d).requires_grad_(True)
    out = MultiScaleDeformableAttnFunction.apply(v, shapes.to(d), starts.to(d), lc, aw, 64)
    ref = oracle.msda_fwd(value, shapes, starts, loc, attw)
    assert torch.allclose(out.detach().cpu(), ref, rtol=1e-5, atol=1e-5)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(9))
    out.backward(g.to(d))
    gv, gl, ga = oracle.msda_bwd(value, shapes, starts, loc, attw, g)
    assert torch.allclose(v.grad.cpu(), gv, rtol=1e-4, atol=1e-4)
    assert torch.allclose(aw.grad.cpu(), ga, rtol=1e-4, atol=1e-5)
    assert torch.allclose(lc.grad.cpu(), gl, rtol=1e-3, atol=1e-4)


def test_msda_empty_and_errors(hip):
    from selfocc_amd._lib import SelfOccHipError
    d = torch.device("cuda:0")
    value, shapes, starts, loc, attw = make_case(1, 4, 2, 8, [[5, 4]], 3)
    out = MultiScaleDeformableAttnFunction.apply(value.to(d), shapes.to(d), starts.to(d), loc[:, :0].to(d), attw[:, :0].to(d), 64)
    assert out.shape == (1, 0, 16)
    bad = torch.randn(1, 20, 2, 6)  # d = 6 unsupported
    with pytest.raises(SelfOccHipError):
        MultiScaleDeformableAttnFunction.apply(bad.to(d), shapes.to(d), starts.to(d),
                                               loc.to(d), attw.to(d), 64)
    with pytest.raises(RuntimeError):
        MultiScaleDeformableAttnFunction.apply(value, shapes, starts, loc, attw, 64)  # CPU tensors: no fallback


def test_msda_reference_shapes_properties(hip):
    """nuscenes_occ hw-plane cross-attention shapes (6 cams, 25500 keys, P = 8); linearity
    in value and weights, determinism of forward."""
    d = torch.device("cuda:0")
    shapes = [[48, 100], [24, 50], [12, 25], [6, 13]]
    value, sh, st, loc, attw = make_case(6, 20000, 6, 16, shapes, 8, seed=2)
    v, lc, aw = value.to(d), loc.to(d), attw.to(d)
    f = lambda V, A: MultiScaleDeformableAttnFunction.apply(V, sh.to(d), st.to(d), lc, A, 64)
    o1, o2 = f(v, aw), f(v, aw)
    assert torch.equal(o1, o2)
    assert torch.allclose(f(2 * v, aw), 2 * o1, rtol=1e-5, atol=1e-5)
    assert torch.allclose(f(v, 0.5 * aw), 0.5 * o1, rtol=1e-5, atol=1e-5)
    sub = oracle.msda_fwd(value[:1, :, :, :], sh, st, loc[:1, :500], attw[:1, :500])
    assert torch.allclose(o1[:1, :500].cpu(), sub, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", [
    (2, 700, 6, 16, [[24, 50], [12, 25], [6, 13], [3, 7]], 8),     # several bands on level 0, one on the rest
    (1, 3000, 2, 16, [[40, 60], [5, 7]], 8),                       # many bands + query chunks on the 5x7 level
    (1, 900, 3, 32, [[30, 40], [15, 20]], 4),                      # 32 channels: 2 points per atomic instruction
    (2, 500, 2, 8, [[30, 40], [15, 20], [8, 10]], 5),              # 8 channels, odd P (keys not 16-byte aligned)
    (1, 400, 2, 4, [[64, 100]], 3),                                # 4 channels, single level
])
def test_msda_bwd_banded_decompositions(hip, case):
    """banded backward (bands x query chunks x channel widths) == oracle == atomic kernel"""
    import selfocc_amd.msda as M
    value, shapes, starts, loc, attw = make_case(*case, seed=11)
    # spread the points beyond the map on every side so that 'outside' keys and partial corners occur
    loc = (loc - 0.5) * 1.3 + 0.5
    d = torch.device("cuda:0")
    grads = {}
    for mode in ("banded", "atomic"):
        M.BACKWARD_MODE = mode
        try:
            v = value.to(d).requires_grad_(True); lc = loc.to(d).requires_grad_(True); aw = attw.to(d).requires_grad_(True)
            out = MultiScaleDeformableAttnFunction.apply(v, shapes.to(d), starts.to(d), lc, aw, 64)
            g = torch.randn(out.shape, generator=torch.Generator().manual_seed(9))
            out.backward(g.to(d))
            grads[mode] = (v.grad.cpu(), lc.grad.cpu(), aw.grad.cpu())
        finally:
            M.BACKWARD_MODE = "banded"
    gv, gl, ga = oracle.msda_bwd(value, shapes, starts, loc, attw, g)
    for mode, (v_g, l_g, a_g) in grads.items():
        assert torch.allclose(v_g, gv, rtol=1e-4, atol=2e-4), mode
        assert torch.allclose(a_g, ga, rtol=1e-4, atol=1e-5), mode
        assert torch.allclose(l_g, gl, rtol=1e-3, atol=1e-4), mode
    # f64 accumulation inside a band: run-to-run differences only from the float flush of query chunks
    M.BACKWARD_MODE = "banded"
    v = value.to(d).requires_grad_(True)
    out = MultiScaleDeformableAttnFunction.apply(v, shapes.to(d), starts.to(d), loc.to(d), attw.to(d), 64)
    out.backward(g.to(d))
    assert torch.allclose(v.grad.cpu(), grads["banded"][0], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("kind,P,L", [(0, 4, 3), (1, 8, 4), (1, 48, 4), (2, 12, 3), (2, 70, 3)])
def test_msda_fused_prologue_matches_unfused(hip, kind, P, L):
    """selfocc_msda_fused_fwd (softmax + ref + off / (W, H) in-kernel) == torch prologue + plain op"""
    from selfocc_amd.msda import msda_fused_inference
    g = torch.Generator().manual_seed(kind * 100 + P)
    shapes = torch.tensor([[24, 50], [12, 25], [6, 13], [3, 7]][:L])
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    bs, nq, H, D = 2, 137, 6, 16
    value = torch.randn(bs, nv, H, D, generator=g)
    off = torch.randn(bs, nq, H, L, P, 2, generator=g) * 3
    logits = torch.randn(bs, nq, H, L * P, generator=g) * 2
    ref = torch.rand(*{0: (bs, nq, L, 2), 1: (bs, nq, P, 2), 2: (bs, nq, L, P, 2)}[kind], generator=g)
    d = torch.device("cuda:0")
    got = msda_fused_inference(value.to(d), shapes.to(d), starts.to(d), ref.to(d), kind, off.to(d), logits.to(d))
    normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1).float()
    r = ref[:, :, None, :, None, :] if kind == 0 else (ref[:, :, None, None, :, :] if kind == 1 else ref[:, :, None, :, :, :])
    loc = r + off / normalizer[None, None, None, :, None, :]
    aw = logits.softmax(-1).view(bs, nq, H, L, P)
    want = oracle.msda_fwd(value, shapes, starts, loc.contiguous(), aw.contiguous())
    assert torch.allclose(got.cpu(), want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("P,L,D,cams", [(8, 4, 16, 5), (48, 4, 16, 5), (5, 2, 8, 5), (70, 3, 16, 5), (3, 1, 32, 5),
                                        (8, 4, 16, 1), (48, 4, 16, 1)])
def test_msda_cross_camera_loop_matches_per_camera_ops(hip, P, L, D, cams):
    """selfocc_msda_cross_fwd == sum over the visible cameras of the plain op (oracle-checked above)
    / max(#visible, 1): queries seen by no camera, one camera and all cameras; cams = 1 is the mono SemanticKITTI /
    KITTI-raw configs (BASELINE configs[3])."""
    from selfocc_amd.msda import msda_cross_inference
    g = torch.Generator().manual_seed(P * 10 + L)
    shapes = torch.tensor([[24, 50], [12, 25], [6, 13], [3, 7]][:L])
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    nq, H = 203, 3
    value = torch.randn(cams, nv, H, D, generator=g)
    off = torch.randn(nq, H, L, P, 2, generator=g) * 3
    logits = torch.randn(nq, H, L * P, generator=g) * 2
    ref = torch.rand(cams, nq, P, 2, generator=g) * 1.4 - 0.2
    vis = torch.rand(cams, nq, generator=g) < 0.4
    vis[:, 0] = False; vis[:, 1] = True; vis[:, 2] = False; vis[min(3, cams - 1), 2] = True
    d = torch.device("cuda:0")
    got = msda_cross_inference(value.to(d), shapes.to(d), starts.to(d), ref.to(d), vis.to(d), off.to(d), logits.to(d)).cpu()
    aw = logits.softmax(-1).view(nq, H, L, P)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    want = torch.zeros(nq, H * D)
    for c in range(cams):
        loc = ref[c][:, None, None, :, :] + off / norm[None, None, :, None, :]
        o = oracle.msda_fwd(value[c:c + 1], shapes, starts, loc[None], aw[None])[0]
        want += o * vis[c][:, None]
    want = want / vis.sum(0).clamp(min=1)[:, None]
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (got - want).abs().max()
    assert torch.all(got[0] == 0)


@pytest.mark.parametrize("kind,P,L,D", [(1, 8, 4, 16), (1, 48, 4, 16), (2, 12, 3, 16), (0, 4, 3, 16), (2, 70, 3, 8), (1, 5, 2, 32)])
def test_msda_fused_training_matches_unfused_autograd(hip, kind, P, L, D):
    """MSDAFusedFunction (prologue fused in forward AND backward) == torch softmax / loc + plain op under
    autograd: output and the gradients w.r.t. value, raw offsets and raw logits."""
    from selfocc_amd.msda import MSDAFusedFunction, msda_fused_supported
    g = torch.Generator().manual_seed(kind * 100 + P + D)
    shapes = torch.tensor([[24, 50], [12, 25], [6, 13], [3, 7]][:L])
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    bs, nq, H = 2, 301, 3
    host = [int(v) for v in shapes.reshape(-1)]
    assert msda_fused_supported(host, bs, nq, H, D, L, P)
    d = torch.device("cuda:0")
    value = torch.randn(bs, nv, H, D, generator=g).to(d)
    off = (torch.randn(bs, nq, H, L, P, 2, generator=g) * 3).to(d)
    logits = (torch.randn(bs, nq, H, L * P, generator=g) * 2).to(d)
    ref = (torch.rand(*{0: (bs, nq, L, 2), 1: (bs, nq, P, 2), 2: (bs, nq, L, P, 2)}[kind], generator=g) * 1.3 - 0.15).to(d)
    gout = torch.randn(bs, nq, H * D, generator=g).to(d)

    def run(fused):
        v, o, lg = (t.clone().requires_grad_(True) for t in (value, off, logits))
        if fused:
            out = MSDAFusedFunction.apply(v, shapes.to(d), starts.to(d), ref, kind, o, lg, host)
        else:
            aw = lg.softmax(-1).view(bs, nq, H, L, P)
            norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float().to(d)
            r = {0: ref[:, :, None, :, None, :] if kind == 0 else None, 1: ref[:, :, None, None, :, :] if kind == 1 else None,
                 2: ref[:, :, None, :, :, :] if kind == 2 else None}[kind]
            loc = r + o / norm[None, None, None, :, None, :]
            out = MultiScaleDeformableAttnFunction.apply(v, shapes.to(d), starts.to(d), loc, aw, 64)
        out.backward(gout)
        return out.detach(), v.grad, o.grad, lg.grad

    a, b = run(True), run(False)
    assert torch.allclose(a[0], b[0], rtol=1e-4, atol=1e-5)
    assert torch.allclose(a[1], b[1], rtol=1e-4, atol=2e-4)                      # grad value
    assert torch.allclose(a[2], b[2], rtol=1e-3, atol=1e-4 * b[2].abs().max().item())   # grad offsets
    assert torch.allclose(a[3], b[3], rtol=1e-3, atol=1e-4 * b[3].abs().max().item())   # grad logits


@pytest.mark.parametrize("P,L,D,cams", [(8, 4, 16, 4), (48, 4, 16, 4), (5, 2, 8, 4), (3, 1, 32, 4), (8, 4, 16, 1), (48, 4, 16, 1)])
def test_msda_cross_training_matches_per_camera_autograd(hip, P, L, D, cams):
    """MSDACrossFunction (camera loop, fused prologue, both directions) == mean over the visible cameras of
    the plain op under torch autograd: output and gradients w.r.t. value, raw offsets, raw logits."""
    from selfocc_amd.msda import MSDACrossFunction, msda_fused_supported
    g = torch.Generator().manual_seed(P * 7 + L)
    shapes = torch.tensor([[24, 50], [12, 25], [6, 13], [3, 7]][:L])
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    nq, H = 260, 3             # cams = 1: the mono KITTI configs
    host = [int(v) for v in shapes.reshape(-1)]
    assert msda_fused_supported(host, cams, nq, H, D, L, P)
    d = torch.device("cuda:0")
    value = torch.randn(cams, nv, H, D, generator=g).to(d)
    off = (torch.randn(nq, H, L, P, 2, generator=g) * 3).to(d)
    logits = (torch.randn(nq, H, L * P, generator=g) * 2).to(d)
    ref = (torch.rand(cams, nq, P, 2, generator=g) * 1.4 - 0.2).to(d)
    vis = torch.rand(cams, nq, generator=g) < 0.45
    vis[:, 0] = False; vis[:, 1] = True
    vis = vis.to(d)
    gout = torch.randn(nq, H * D, generator=g).to(d)

    def run(fused):
        v, o, lg = (t.clone().requires_grad_(True) for t in (value, off, logits))
        if fused:
            out = MSDACrossFunction.apply(v, shapes.to(d), starts.to(d), ref, vis, o, lg, host)
        else:
            aw = lg.softmax(-1).view(1, nq, H, L, P)
            norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float().to(d)
            out = torch.zeros(nq, H * D, device=d)
            for c in range(cams):
                loc = ref[c][None, :, None, None, :, :] + o[None] / norm[None, None, None, :, None, :]
                oc = MultiScaleDeformableAttnFunction.apply(v[c:c + 1], shapes.to(d), starts.to(d), loc, aw, 64)[0]
                out = out + oc * vis[c][:, None]
            out = out / vis.sum(0).clamp(min=1)[:, None]
        out.backward(gout)
        return out.detach(), v.grad, o.grad, lg.grad

    a, b = run(True), run(False)
    assert torch.allclose(a[0], b[0], rtol=1e-4, atol=1e-5)
    assert torch.allclose(a[1], b[1], rtol=1e-4, atol=2e-4)
    assert torch.allclose(a[2], b[2], rtol=1e-3, atol=1e-4 * b[2].abs().max().item())
    assert torch.allclose(a[3], b[3], rtol=1e-3, atol=1e-4 * b[3].abs().max().item())


def _full_size_case():
    """nuscenes_occ hw-plane cross-attention at FULL size: 6 cams x 22016 queries x 6 heads x 4 levels x 8 points = 25.4 M
    points on FPN maps 96x200 .. 12x25, spatially coherent sampling locations"""
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(3)
    bs, nq, H, D, P = 6, 22016, 6, 16, 8
    shapes = torch.tensor([[96, 200], [48, 100], [24, 50], [12, 25]])
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum()); L = 4
    value = torch.randn(bs, nv, H, D, device=d, generator=g)
    side = int(nq ** 0.5) + 1
    qi = torch.arange(nq, device=d)
    base = torch.stack([(qi % side) / side, (qi // side) / side], -1)
    loc = base[None, :, None, None, None, :] + torch.randn(bs, nq, H, L, P, 2, device=d, generator=g) * 0.03
    attw = torch.softmax(torch.randn(bs, nq, H, L * P, device=d, generator=g), -1).view(bs, nq, H, L, P)
    gout = torch.randn(bs, nq, H * D, device=d, generator=g)
    return d, shapes, starts, value, loc, attw, gout


def test_msda_full_size_forward_backward_vs_oracle(hip):
    """EVERY element of the full-size case against the C oracle (forward on all host threads, backward single-threaded,
    ~10 s): output, grad_loc, grad_attw to float rounding; grad_value (up to ~2300 float adds per element on the 12x25
    level, summed in a different order on each side) to 2e-4 of its scale."""
    import oracle
    d, shapes, starts, value, loc, attw, gout = _full_size_case()
    v, lc, aw = (t.clone().requires_grad_(True) for t in (value, loc, attw))
    out = MultiScaleDeformableAttnFunction.apply(v, shapes.to(d), starts.to(d), lc, aw, 64)
    out.backward(gout)
    want = oracle.msda_fwd(value.cpu(), shapes, starts, loc.cpu(), attw.cpu())
    assert torch.allclose(out.detach().cpu(), want, rtol=1e-4, atol=1e-5)
    gv, gl, ga = oracle.msda_bwd(value.cpu(), shapes, starts, loc.cpu(), attw.cpu(), gout.cpu())
    assert torch.allclose(aw.grad.cpu(), ga, rtol=1e-4, atol=1e-5)
    # grad_loc = (corner differences) x map size: up to ~200 x the value scale; relative to that scale
    assert (lc.grad.cpu() - gl).abs().max().item() < 1e-4 * gl.abs().max().item()
    assert (v.grad.cpu() - gv).abs().max().item() < 2e-4 * gv.abs().max().item()


def test_msda_backward_full_size_banded_vs_atomic(hip):
    """the full-size case: the banded LDS-f64 backward and the global-atomic backward agree;
    grad_attw / grad_loc (no atomics in either) to float rounding, grad_value to the float-atomic noise."""
    import selfocc_amd.msda as M
    d, shapes, starts, value, loc, attw, gout = _full_size_case()
    res = {}
    for mode in ("banded", "atomic"):
        M.BACKWARD_MODE = mode
        try:
            v, lc, aw = (t.clone().requires_grad_(True) for t in (value, loc, attw))
            out = MultiScaleDeformableAttnFunction.apply(v, shapes.to(d), starts.to(d), lc, aw, 64)
            out.backward(gout)
            res[mode] = (v.grad, lc.grad, aw.grad)
        finally:
            M.BACKWARD_MODE = "banded"
    (gv_b, gl_b, ga_b), (gv_a, gl_a, ga_a) = res["banded"], res["atomic"]
    assert torch.allclose(ga_b, ga_a, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gl_b, gl_a, rtol=1e-3, atol=1e-3)
    scale = gv_a.abs().max().item()
    assert (gv_b - gv_a).abs().max().item() < 2e-4 * scale          # ~2300 float adds per element on the 12x25 level
    assert torch.isfinite(gv_b).all() and gv_b.abs().sum() > 0


@pytest.mark.parametrize("P,L,D", [(8, 4, 16), (48, 4, 16), (5, 2, 8), (3, 1, 32)])
def test_msda_head_major_value_layout(hip, P, L, D):
    """value_layout = SO_VALUE_HEAD_MAJOR, (bs, heads, nv, d): the fused and camera-loop ops return the SAME output
    bits as with mmcv's (bs, nv, heads, d) (the layout only moves where the corners are read from), and the same
    gradients, g_value in the layout value came in."""
    from selfocc_amd.msda import (MSDAFusedFunction, MSDACrossFunction, msda_fused_inference, msda_cross_inference,
                                  to_head_major)
    g = torch.Generator().manual_seed(P * 11 + L)
    shapes = torch.tensor([[24, 50], [12, 25], [6, 13], [3, 7]][:L])
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    cams, nq, H = 3, 260, 3
    host = [int(v) for v in shapes.reshape(-1)]
    d = torch.device("cuda:0")
    sh, st = shapes.to(d), starts.to(d)
    value = torch.randn(cams, nv, H, D, generator=g).to(d)
    value_hm = to_head_major(value)
    assert value_hm.shape == (cams, H, nv, D) and value_hm.is_contiguous()

    # fused (self-attention form), reference kind 1
    off = (torch.randn(cams, nq, H, L, P, 2, generator=g) * 3).to(d)
    logits = (torch.randn(cams, nq, H, L * P, generator=g) * 2).to(d)
    ref = (torch.rand(cams, nq, P, 2, generator=g) * 1.3 - 0.15).to(d)
    a = msda_fused_inference(value, sh, st, ref, 1, off, logits)
    b = msda_fused_inference(value_hm, sh, st, ref, 1, off, logits, head_major=True)
    assert torch.equal(a, b)
    gout = torch.randn(cams, nq, H * D, generator=g).to(d)

    def run_fused(hm):
        v, o, lg = (t.clone().requires_grad_(True) for t in (value_hm if hm else value, off, logits))
        MSDAFusedFunction.apply(v, sh, st, ref, 1, o, lg, host, hm).backward(gout)
        return (v.grad.permute(0, 2, 1, 3) if hm else v.grad), o.grad, lg.grad
    for x, y in zip(run_fused(True), run_fused(False)):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-5 * y.abs().max().item())

    # camera loop
    offc, lgc = off[0].contiguous(), logits[0].contiguous()
    refc = (torch.rand(cams, nq, P, 2, generator=g) * 1.4 - 0.2).to(d)
    vis = (torch.rand(cams, nq, generator=g) < 0.5).to(d)
    a = msda_cross_inference(value, sh, st, refc, vis, offc, lgc)
    b = msda_cross_inference(value_hm, sh, st, refc, vis, offc, lgc, head_major=True)
    assert torch.equal(a, b)
    goutc = torch.randn(nq, H * D, generator=g).to(d)

    def run_cross(hm):
        v, o, lg = (t.clone().requires_grad_(True) for t in (value_hm if hm else value, offc, lgc))
        MSDACrossFunction.apply(v, sh, st, refc, vis, o, lg, host, hm).backward(goutc)
        return (v.grad.permute(0, 2, 1, 3) if hm else v.grad), o.grad, lg.grad
    for x, y in zip(run_cross(True), run_cross(False)):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-5 * y.abs().max().item())


@pytest.mark.parametrize("P,L,D", [(8, 4, 16), (48, 4, 16), (12, 3, 16)])     # bf16 `value` is built for 16 channels per head (the shipped lifters)
def test_msda_bf16_value_storage(hip, P, L, D):
    """value_dtype = SO_DTYPE_BF16 (bfloat16 STORAGE of value, float32 arithmetic): the fused and camera-loop ops give
    exactly what the float32 kernels give on bf16-rounded values (forward: same bits; backward: same gradients, g_value
    float32), and stay within the bf16 rounding of value (2^-8 relative, stated tolerance) of the float32 result."""
    from selfocc_amd.msda import MSDAFusedFunction, MSDACrossFunction, msda_fused_inference, msda_cross_inference
    g = torch.Generator().manual_seed(P * 13 + L)
    shapes = torch.tensor([[24, 50], [12, 25], [6, 13], [3, 7]][:L])
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    cams, nq, H = 3, 260, 3
    host = [int(v) for v in shapes.reshape(-1)]
    d = torch.device("cuda:0")
    sh, st = shapes.to(d), starts.to(d)
    value = torch.randn(cams, nv, H, D, generator=g).to(d)
    v_bf = value.to(torch.bfloat16)
    v_rt = v_bf.float()                                     # what the bf16 kernels see, as float32
    off = (torch.randn(cams, nq, H, L, P, 2, generator=g) * 3).to(d)
    logits = (torch.randn(cams, nq, H, L * P, generator=g) * 2).to(d)
    ref = (torch.rand(cams, nq, P, 2, generator=g) * 1.3 - 0.15).to(d)
    a = msda_fused_inference(v_bf, sh, st, ref, 1, off, logits)
    b = msda_fused_inference(v_rt, sh, st, ref, 1, off, logits)
    full = msda_fused_inference(value, sh, st, ref, 1, off, logits)
    assert a.dtype == torch.float32 and torch.equal(a, b)
    assert (a - full).abs().max().item() <= 2.0 ** -8 * value.abs().max().item()
    gout = torch.randn(cams, nq, H * D, generator=g).to(d)

    def run_fused(bf):
        v, o, lg = ((value if bf else v_rt).clone().requires_grad_(True), off.clone().requires_grad_(True),
                    logits.clone().requires_grad_(True))
        MSDAFusedFunction.apply(v, sh, st, ref, 1, o, lg, host, False, bf).backward(gout)
        assert v.grad.dtype == torch.float32
        return v.grad, o.grad, lg.grad
    for x, y in zip(run_fused(True), run_fused(False)):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-5 * y.abs().max().item())

    offc, lgc = off[0].contiguous(), logits[0].contiguous()
    refc = (torch.rand(cams, nq, P, 2, generator=g) * 1.4 - 0.2).to(d)
    vis = (torch.rand(cams, nq, generator=g) < 0.5).to(d)
    a = msda_cross_inference(v_bf, sh, st, refc, vis, offc, lgc)
    b = msda_cross_inference(v_rt, sh, st, refc, vis, offc, lgc)
    assert torch.equal(a, b)
    goutc = torch.randn(nq, H * D, generator=g).to(d)

    def run_cross(bf):
        v, o, lg = ((value if bf else v_rt).clone().requires_grad_(True), offc.clone().requires_grad_(True),
                    lgc.clone().requires_grad_(True))
        MSDACrossFunction.apply(v, sh, st, refc, vis, o, lg, host, False, bf).backward(goutc)
        return v.grad, o.grad, lg.grad
    for x, y in zip(run_cross(True), run_cross(False)):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-5 * y.abs().max().item())


def test_msda_kernels_beside_a_bf16_mfma_kernel_on_another_stream(hip):
    """The shipped-size camera-loop, fused and plain MSDA kernels stay BITWISE repeatable while selfocc_linear_fwd's bf16 x 3 MFMA
    kernel runs on a second HIP stream.  Round 5 (profiles/r5_b_packed_fp32_mfma.txt): with compiler-formed packed FP32
    (`v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` in the bilinear setup) every such launch had a few wrong (query, head) rows —
    the low half of the packed product came back as if lx * W were 0 — whenever a v_mfma_f32_16x16x32_bf16 wave shared the SIMD:
    a second stream, a second process on the GPU (the two-rank tests), or both phases inside one kernel.  csrc/build.sh now
    builds without the vectorizers; a build that brings the instruction form back fails here (and in tests/test_isa_lint.py)."""
    from selfocc_amd.msda import msda_cross_inference, msda_fused_inference, multi_scale_deformable_attn, to_head_major
    from selfocc_amd.linear import linear_fwd
    d0 = torch.device("cuda:0")
    g = torch.Generator(device=d0).manual_seed(11)
    heads, d, K, cams, nq = 6, 16, 96, 6, 66049
    shapes = torch.tensor([[96, 200], [48, 100], [24, 50], [12, 25]])
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    sh, st = shapes.to(d0), starts.to(d0)
    value = torch.randn(cams, nv, heads, d, device=d0, generator=g)
    v_hm = to_head_major(value)
    L, P = 4, 8
    off = torch.randn(nq, heads, L, P, 2, device=d0, generator=g) * 2
    logits = torch.randn(nq, heads, L * P, device=d0, generator=g)
    vis = torch.rand(cams, nq, device=d0, generator=g) < 0.35
    refc = torch.rand(cams, nq, P, 2, device=d0, generator=g) * 1.2 - 0.1
    ref1 = torch.rand(1, nq, P, 2, device=d0, generator=g) * 1.2 - 0.1
    loc = torch.rand(1, 22016, heads, L, P, 2, device=d0, generator=g) * 1.1 - 0.05
    aw = torch.softmax(torch.randn(1, 22016, heads, L * P, device=d0, generator=g), -1).view(1, 22016, heads, L, P)
    victims = {
        "msda_cross_fwd": lambda: msda_cross_inference(v_hm, sh, st, refc, vis, off, logits, True),
        "msda_fused_fwd": lambda: msda_fused_inference(v_hm[:1], sh, st, ref1, 1, off[None], logits[None], True),
        "msda_fwd": lambda: multi_scale_deformable_attn(value[:1], sh, st, loc, aw),
    }
    x = torch.randn(78899, K, device=d0, generator=g)
    w = torch.randn(432, K, device=d0, generator=g) * 0.1
    b = torch.randn(432, device=d0, generator=g)
    with torch.no_grad():
        quiet = {k: f().clone() for k, f in victims.items()}
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        bad = {k: 0 for k in victims}
        for k, f in victims.items():
            for _ in range(40):
                with torch.cuda.stream(sa):
                    linear_fwd(x, w, b)
                with torch.cuda.stream(sb):
                    bad[k] += not torch.equal(f(), quiet[k])
            torch.cuda.synchronize()
    assert not any(bad.values()), f"launches (of 40) that differ from the quiet result: {bad}"


@pytest.mark.parametrize("L,P,nq,cams", [(4, 8, 3001, 3), (3, 12, 1777, 0), (4, 48, 501, 2)])
def test_merged_offset_logit_rows_equal_the_dense_pair(hip, L, P, nq, cams):
    """ABI 32 ``ol_stride``: the fused / camera-loop kernels read the raw offsets and logits of a query from ONE merged row
    [heads*L*P*2 offsets | heads*L*P logits] (the stacked sampling_offsets | attention_weights projection) and write the
    gradient of that row — bit for bit what the dense pair gives, in both directions."""
    from selfocc_amd.msda import (msda_fused_inference, msda_cross_inference, MSDAFusedFunction, MSDACrossFunction, to_head_major)
    d0 = torch.device("cuda:0")
    g = torch.Generator(device=d0).manual_seed(L * 100 + P)
    heads, d = 6, 16
    shapes = torch.tensor([[24, 50], [12, 25], [6, 13], [3, 7]][:L])
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    sh, st = shapes.to(d0), starts.to(d0)
    host = [int(v) for v in shapes.reshape(-1).tolist()]
    nb = max(cams, 1)
    v_hm = to_head_major(torch.randn(nb, nv, heads, d, device=d0, generator=g))
    off = torch.randn(nq, heads, L, P, 2, device=d0, generator=g) * 2
    lg = torch.randn(nq, heads, L * P, device=d0, generator=g)
    ol = torch.cat([off.reshape(nq, -1), lg.reshape(nq, -1)], -1).contiguous()
    gout = torch.randn(nq, heads * d, device=d0, generator=g)
    if cams:
        ref = torch.rand(cams, nq, P, 2, device=d0, generator=g) * 1.2 - 0.1
        vis = torch.rand(cams, nq, device=d0, generator=g) < 0.5
        a = msda_cross_inference(v_hm, sh, st, ref, vis, off, lg, True)
        b = msda_cross_inference(v_hm, sh, st, ref, vis, ol, None, True, (L, P))
        assert torch.equal(a, b)

        def run(merged):
            v = v_hm.clone().requires_grad_(True)
            if merged:
                x = ol.clone().requires_grad_(True)
                MSDACrossFunction.apply(v, sh, st, ref, vis, x, None, host, True, False, (L, P)).backward(gout)
                return v.grad, x.grad
            o, l_ = off.clone().requires_grad_(True), lg.clone().requires_grad_(True)
            MSDACrossFunction.apply(v, sh, st, ref, vis, o, l_, host, True, False).backward(gout)
            return v.grad, torch.cat([o.grad.reshape(nq, -1), l_.grad.reshape(nq, -1)], -1)
    else:
        ref = torch.rand(1, nq, L, P, 2, device=d0, generator=g) * 1.1 - 0.05
        a = msda_fused_inference(v_hm, sh, st, ref, 2, off[None], lg[None], True)
        b = msda_fused_inference(v_hm, sh, st, ref, 2, ol[None], None, True, (L, P))
        assert torch.equal(a, b)

        def run(merged):
            v = v_hm.clone().requires_grad_(True)
            if merged:
                x = ol[None].clone().requires_grad_(True)
                MSDAFusedFunction.apply(v, sh, st, ref, 2, x, None, host, True, False, (L, P)).backward(gout[None])
                return v.grad, x.grad[0]
            o, l_ = off[None].clone().requires_grad_(True), lg[None].clone().requires_grad_(True)
            MSDAFusedFunction.apply(v, sh, st, ref, 2, o, l_, host, True, False).backward(gout[None])
            return v.grad, torch.cat([o.grad.reshape(nq, -1), l_.grad.reshape(nq, -1)], -1)
    (gv_m, gol_m), (gv_d, gol_d) = run(True), run(False)
    assert torch.equal(gol_m, gol_d)
    assert torch.allclose(gv_m, gv_d, rtol=1e-5, atol=1e-6 * float(gv_d.abs().max()))        # grad_value: atomic order


def test_merged_offset_logit_projection_module_equals_two_linears(hip, monkeypatch):
    """bricks.merged_off_logits through a whole attention module: outputs and every parameter gradient equal the
    two-Linear route (SELFOCC_MERGED_OFF_LOGITS off) to float32 rounding of the projections' summation order."""
    from selfocc_amd.model import bricks
    from selfocc_amd.registry import MODELS
    import selfocc_amd.model  # noqa: F401
    d0 = torch.device("cuda:0")
    torch.manual_seed(3)
    att = MODELS.build(dict(type='CrossViewHybridAttention', embed_dims=96, num_heads=6, num_levels=3, num_points=12,
                            dropout=0.0, batch_first=True)).to(d0)
    # the bilinear gradient w.r.t. a sampling location is piece-wise constant in the pixel cell, so two correct projections
    # that differ in the last bit can put a location on either side of a pixel edge and move a few gradient entries by O(1)
    # (measured here with random weights: 1e-3 of scale).  Dyadic queries / weights make both projections EXACT, hence equal.
    gi = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for lin, sc in ((att.sampling_offsets, 64.0), (att.attention_weights, 64.0)):
            lin.weight.copy_(torch.randint(-8, 9, lin.weight.shape, generator=gi).float() / sc)
            lin.bias.copy_(torch.randint(-16, 17, lin.bias.shape, generator=gi).float() / 8.0)
    sizes = [(33, 33), (9, 33), (33, 9)]
    shapes = torch.tensor(sizes, device=d0)
    shapes._so_host = [v for s_ in sizes for v in s_]
    starts = torch.tensor([0, 33 * 33, 33 * 33 + 9 * 33], device=d0)
    nq = sum(a * b for a, b in sizes)
    q = (torch.randint(-16, 17, (1, nq, 96), generator=gi).float() / 8.0).to(d0)
    ref = torch.rand(1, nq, 3, 12, 2, device=d0)
    gy = torch.randn(1, nq, 96, device=d0)
    res = {}
    for merged in (True, False):
        monkeypatch.setattr(bricks, 'MERGED_OFF_LOGITS', merged)
        monkeypatch.setattr(bricks, 'LINEAR_FWD_MIN_ROWS', 64)
        att.zero_grad(set_to_none=True)
        qq = q.clone().requires_grad_(True)
        att.train()
        y = att(qq, reference_points=ref, spatial_shapes=shapes, level_start_index=starts)
        y.backward(gy)
        with torch.no_grad():
            att.eval()
            y_inf = att(q, reference_points=ref, spatial_shapes=shapes, level_start_index=starts)
        res[merged] = (y.detach(), y_inf, qq.grad, {n: p.grad.clone() for n, p in att.named_parameters()})
    ym, yim, gqm, gpm = res[True]
    yd, yid, gqd, gpd = res[False]
    err = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)      # of the tensor's scale
    errs = dict(y=err(ym, yd), y_inference=err(yim, yid), g_query=err(gqm, gqd), **{n: err(gpm[n], gpd[n]) for n in gpd})
    bad = {k: v for k, v in errs.items() if v > 1e-5}
    assert not bad, (bad, errs)


def test_abi32_stride_arguments_are_validated_at_the_c_boundary(hip):
    """ol_stride / g_value_stride of the fused entry points (include/selfocc_hip.h, ABI 32): a bad value is an error string,
    never a wild write."""
    import ctypes as C
    from selfocc_amd._lib import lib, ptr, current_stream
    from selfocc_amd import abi
    d0 = torch.device("cuda:0")
    heads, d, L, P, nq, bs = 6, 16, 2, 4, 33, 1
    shapes = torch.tensor([[6, 10], [3, 5]], dtype=torch.int32, device=d0)
    starts = torch.tensor([0, 60], dtype=torch.int32, device=d0)
    nv, LP = 75, L * P
    value = torch.randn(bs, heads, nv, d, device=d0)
    ref = torch.rand(bs, nq, L, 2, device=d0)
    ol = torch.randn(bs, nq, 3 * heads * LP, device=d0)
    out = torch.empty(bs, nq, heads * d, device=d0)
    lg_ptr = C.c_void_p(ol.data_ptr() + 8 * heads * LP)
    st = current_stream(d0)
    l = lib()

    def fwd(ols):
        return l.selfocc_msda_fused_fwd(ptr(value), ptr(shapes), ptr(starts), ptr(ref), 0, ptr(ol), lg_ptr, ptr(out), bs, nv, nq, heads,
                                        d, L, P, 1, abi.DTYPE_F32, ols, st)
    assert fwd(3 * heads * LP) == 0
    for bad in (3 * heads * LP - 2, 3 * heads * LP + 1, -4):
        assert fwd(bad) != 0 and b"ol_stride" in l.selfocc_last_error()
    host = (C.c_int32 * 4)(6, 10, 3, 5)
    g_out = torch.randn(bs, nq, heads * d, device=d0)
    g_ol = torch.empty_like(ol)
    g_rows = torch.zeros(bs, nv, 2, heads, d, device=d0)
    nbytes = int(l.selfocc_msda_bwd_banded_workspace(bs, nq, heads, L, P))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d0)

    def bwd(gvs):
        return l.selfocc_msda_fused_bwd(ptr(value), ptr(shapes), ptr(starts), C.cast(host, C.c_void_p), ptr(ref), 0, ptr(ol), lg_ptr,
                                        ptr(g_out), ptr(g_rows), ptr(g_ol), C.c_void_p(g_ol.data_ptr() + 8 * heads * LP), bs, nv, nq,
                                        heads, d, L, P, 1, abi.DTYPE_F32, 3 * heads * LP, gvs, ptr(ws), nbytes, st)
    assert bwd(2 * heads * d) == 0
    torch.cuda.synchronize()
    assert float(g_rows[:, :, 0].abs().sum()) > 0 and float(g_rows[:, :, 1].abs().sum()) == 0      # only this op's column block is written
    for bad in (heads * d - 16, -1):
        assert bwd(bad) != 0 and b"g_value_stride" in l.selfocc_last_error()
