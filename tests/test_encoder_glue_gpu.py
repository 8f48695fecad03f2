"""GPU: the training-path glue of the encoders (round 3) against the torch formulation the reference runs under autograd.

  * selfocc_flatten_feats == 2 broadcast adds per level + cat + permute (tpvformer_encoder.py:261-277), bit for bit;
    _FlattenFeats' hand-written backward == autograd through those ops;
  * _TallLinearHeadsMulti (one merged value projection per layer) == three nn.Linear + head-major transposes;
  * _TallLinearReLU == nn.Linear -> nn.ReLU(inplace=True) (mmcv FFN's first layer).
"""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
D0 = torch.device("cuda:0")


def _feats(B, N, C, shapes, seed=0, requires_grad=False):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, N, C, h, w, generator=g).to(D0).requires_grad_(requires_grad) for h, w in shapes]


@pytest.mark.parametrize("B,N,C,shapes", [(1, 6, 96, [(96, 200), (48, 100), (24, 50), (12, 25)]),
                                          (2, 3, 32, [(7, 9), (3, 5)]), (1, 1, 128, [(1, 1)]), (1, 2, 20, [(5, 13)] * 8)])
def test_flatten_feats_hip_is_the_torch_formula_bit_for_bit(hip, B, N, C, shapes):
    from selfocc_amd.model.encoder import tpvformer as T
    feats = _feats(B, N, C, shapes)
    cams, lvls = torch.randn(N, C, device=D0), torch.randn(len(shapes) + 1, C, device=D0)
    with torch.no_grad():
        got = T._flatten_feats(cams, lvls, feats)
        want = T._flatten_feats_torch(cams, lvls, feats)
    assert got.shape == want.shape == (N, sum(h * w for h, w in shapes), B, C)
    assert got.is_contiguous() and torch.equal(got, want)


def test_flatten_feats_rejects_bad_arguments(hip):
    import ctypes as C
    from selfocc_amd._lib import lib
    assert lib().selfocc_flatten_feats(None, None, 0, 1, 1, 1, None, None, None, None) != 0
    assert b"levels" in lib().selfocc_last_error()


def test_flatten_feats_backward_vs_autograd(hip):
    from selfocc_amd.model.encoder import tpvformer as T
    shapes = [(24, 50), (12, 25), (5, 7)]
    cams = torch.randn(3, 32, device=D0, requires_grad=True)
    lvls = torch.randn(4, 32, device=D0, requires_grad=True)        # one level more than maps: its gradient row stays zero
    fa, fb = _feats(2, 3, 32, shapes, 1, True), _feats(2, 3, 32, shapes, 1, True)
    ya = T._FlattenFeats.apply(cams, lvls, *fa)
    g = torch.randn_like(ya)
    ga = torch.autograd.grad(ya, [cams, lvls] + fa, g)
    yb = T._flatten_feats_torch(cams, lvls, fb)
    gb = torch.autograd.grad(yb, [cams, lvls] + fb, g)
    assert torch.equal(ya, yb)
    for a, b in zip(ga, gb):
        assert a.shape == b.shape
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * b.abs().max().item())
    assert torch.all(ga[1][3] == 0)


def test_merged_value_projection_vs_three_linears(hip):
    from selfocc_amd.model import bricks
    torch.manual_seed(0)
    cams, nv, K = 6, 2600, 96
    lins = [nn.Linear(K, 96).to(D0) for _ in range(3)]
    x = torch.randn(cams * nv, K, device=D0, requires_grad=True)
    vs = bricks.value_proj_head_major_multi(lins, x, nv, 6)
    assert vs is not None and len(vs) == 3 and vs[0].shape == (cams, 6, nv, 16)
    gs = [torch.randn_like(v) for v in vs]
    gs[1] = None                                                   # a plane whose result nobody used
    torch.autograd.backward([v for v, g in zip(vs, gs) if g is not None], [g for g in gs if g is not None])
    got = [x.grad.clone()] + [p.grad.clone() for l in lins for p in (l.weight, l.bias)]
    x.grad = None
    for l in lins:
        l.weight.grad = l.bias.grad = None
    want_vs = [l(x).view(cams, nv, 6, 16).permute(0, 2, 1, 3) for l in lins]
    for v, w in zip(vs, want_vs):
        assert torch.allclose(v, w, rtol=1e-5, atol=1e-5)
    torch.autograd.backward([w for w, g in zip(want_vs, gs) if g is not None], [g for g in gs if g is not None])
    want = [x.grad] + [p.grad if p.grad is not None else torch.zeros_like(p) for l in lins for p in (l.weight, l.bias)]
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * max(b.abs().max().item(), 1e-3))


def test_ffn_fused_relu_vs_sequential(hip, monkeypatch):
    from selfocc_amd.model import bricks
    torch.manual_seed(0)
    ffn = bricks.FFN(embed_dims=96, feedforward_channels=192, ffn_drop=0.0).to(D0).train()
    x = torch.randn(1, 5000, 96, device=D0)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(bricks, "FUSED_FFN_RELU", fused)
        xi = x.clone().requires_grad_(True)
        y = ffn(xi)
        y.square().sum().backward()
        outs.append([y.detach(), xi.grad] + [p.grad.clone() for p in ffn.parameters()])
        for p in ffn.parameters():
            p.grad = None
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * b.abs().max().item())


def test_cached_positional_encodings_follow_the_parameters(hip):
    """eval frames reuse the planes' positional encodings; an in-place parameter update (optimizer step, load_state_dict)
    invalidates them"""
    import test_sync_free_gpu as S
    th, lifter, enc, head, _ = S._stages(train=False)
    metas, feats, _ = S._frame(th, 0)
    with torch.no_grad():
        torch.manual_seed(0)
        for name, p in enc.named_parameters():      # the offset / weight projections start at zero weight (mmcv init):
            if 'sampling_offsets.weight' in name or 'attention_weights.weight' in name:     # make them see the query
                p.normal_(std=0.05)
        a = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        b = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        assert all(torch.equal(x, y) for x, y in zip(a, b))
        key0 = enc._pos_cache[0]
        enc.positional_encoding.position_layer_hw.weight.mul_(1.5)
        c = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        assert enc._pos_cache[0] != key0 and not torch.equal(a[0], c[0])
        del enc._pos_cache
        d = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        assert all(torch.equal(x, y) for x, y in zip(c, d))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_training_step_under_autocast(hip, dtype):
    """the reference's `amp` switch (train.py wraps the forward in torch.autocast): the HIP ops compute in float32 under it
    (custom_fwd casts), the torch ops around them in the autocast dtype — one whole training step runs, every gradient is
    finite, and the loss stays close to the float32 step's"""
    import numpy as np
    import test_sync_free_gpu as S

    def step(amp):
        torch.manual_seed(0); np.random.seed(0)
        th, lifter, enc, head, loss_fn = S._stages(train=True)
        for m in (lifter, enc, head):
            for mod in m.modules():
                if isinstance(mod, nn.Dropout):
                    mod.p = 0.0                                       # same arithmetic in both runs
        metas, feats, imgs = S._frame(th, 0)
        with torch.autocast("cuda", dtype=dtype, enabled=amp):
            rep = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
            out = head(rep, metas, global_iter=7)
            total, parts = loss_fn(dict(out, metas=metas, **imgs))
        total.backward()
        params = [p for m in (lifter, enc, head) for p in m.parameters()]
        assert torch.isfinite(total).all() and all(p.grad is None or torch.isfinite(p.grad).all() for p in params)
        assert sum(p.grad is not None for p in params) > 20
        return float(total.detach())
    ref, amp = step(False), step(True)
    assert abs(amp - ref) <= 0.05 * abs(ref) + 1e-3, (amp, ref)


def test_encoder_batch_of_two_runs_both_modes(hip):
    """batch size 2 (the reference trains with 1 per GPU; its masks use batch element 0): the camera-loop kernels step aside
    for the re-batch path, the merged value projection / kept-concatenated planes / flatten kernel take a batch — eval and a
    training step run, finite, and element 0 of an identical pair equals element 1"""
    import copy
    import test_sync_free_gpu as S
    th, lifter, enc, head, _ = S._stages(train=False)
    metas, feats, _ = S._frame(th, 0)
    metas2 = [metas[0], copy.deepcopy(metas[0])]
    feats2 = [f.repeat(2, 1, 1, 1, 1) for f in feats]
    with torch.no_grad():
        rep = enc(lifter(feats2)['representation'], ms_img_feats=feats2, metas=metas2)['representation']
    assert all(r.shape[0] == 2 and torch.isfinite(r).all() for r in rep)
    assert all(torch.allclose(r[0], r[1], rtol=1e-5, atol=1e-5) for r in rep)
    for m in (lifter, enc):
        m.train()
    rep = enc(lifter(feats2)['representation'], ms_img_feats=feats2, metas=metas2)['representation']
    sum(r.square().mean() for r in rep).backward()
    grads = [p.grad for m in (lifter, enc) for p in m.parameters() if p.grad is not None]
    assert len(grads) > 20 and all(torch.isfinite(g).all() for g in grads)


def test_lifter_views_follow_the_parameters(hip):
    """inference: the lifter hands the encoder views of one concatenated tensor (no per-frame copies); an in-place
    parameter update rebuilds it; under autograd the planes are expanded views of the parameters themselves"""
    from selfocc_amd.registry import MODELS
    import selfocc_amd.model  # noqa: F401
    from selfocc_amd.model.encoder import tpvformer as T
    lifter = MODELS.build(dict(type='TPVQueryLifter', tpv_h=5, tpv_w=4, tpv_z=3, dim=8)).to(D0)
    feats = [torch.zeros(1, 2, 8, 2, 2, device=D0)]
    with torch.no_grad():
        a = lifter(feats)['representation']
        assert isinstance(a, T._Planes) and a.cat.shape == (1, 20 + 15 + 12, 8) and T._as_cat(a) is a.cat
        assert all(torch.equal(x, p) for x, p in zip(a, (lifter.tpv_hw, lifter.tpv_zh, lifter.tpv_wz)))
        assert lifter(feats)['representation'].cat is a.cat              # kept
        lifter.tpv_zh.add_(1.0)
        b = lifter(feats)['representation']
        assert b.cat is not a.cat and torch.equal(b[1], lifter.tpv_zh)
    c = lifter(feats)['representation']                                  # autograd: gradients reach the parameters
    sum(x.sum() for x in c).backward()
    assert lifter.tpv_hw.grad is not None and torch.all(lifter.tpv_hw.grad == 1)
    two = lifter([torch.zeros(2, 2, 8, 2, 2, device=D0)])['representation']
    assert two[0].shape == (2, 20, 8)


@pytest.mark.parametrize("n", [7372800 // 16, 1000, 1, 257])
def test_eikonal_loss_hip_vs_torch_formula(hip, n):
    """EikonalLoss through selfocc_eikonal_fwd / _bwd == the reference's torch formula (loss/eikonal_loss.py:19-22) under
    autograd: value and gradient, rows of zero norm included (torch's norm backward gives them a zero gradient)"""
    from selfocc_amd.loss.simple import EikonalLoss
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, 3, generator=g) * 0.7 + 0.3).to(D0)
    x[0] = 0.0
    a = x.clone().requires_grad_(True)
    eik = EikonalLoss()
    la = eik.eikonal(a)
    (la * 0.37).backward()
    b = x.clone().requires_grad_(True)
    lb = ((b.norm(2, dim=-1) - 1) ** 2).mean()
    (lb * 0.37).backward()
    assert torch.allclose(la, lb, rtol=2e-6, atol=1e-8)
    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7 / n) and torch.all(a.grad[0] == 0)
    # deterministic: a fixed order of partial sums
    assert torch.equal(eik.eikonal(x), eik.eikonal(x))


@pytest.mark.parametrize("shape", [(257, 257, 25), (9, 7, 5), (3, 3, 3), (2, 5, 1), (1, 1, 1)])
def test_second_differences_hip_vs_torch_expression(hip, shape):
    """NeuSHead's `second_grad` through selfocc_second_diff_fwd / _bwd == the torch expression it replaces: forward bit for
    bit, backward (gather form) to float rounding of the accumulation order; axes shorter than 3 contribute nothing"""
    from selfocc_amd.model.head.neus_head import _SecondDiff
    g = torch.Generator().manual_seed(sum(shape))
    s0 = torch.randn(*shape, generator=g).to(D0)

    def torch_form(s):
        return torch.cat([(s[2:] - 2 * s[1:-1] + s[:-2]).flatten(), (s[:, 2:] - 2 * s[:, 1:-1] + s[:, :-2]).flatten(),
                          (s[:, :, 2:] - 2 * s[:, :, 1:-1] + s[:, :, :-2]).flatten()])
    a, b = s0.clone().requires_grad_(True), s0.clone().requires_grad_(True)
    ya, yb = _SecondDiff.apply(a), torch_form(b)
    assert ya.shape == yb.shape and torch.equal(ya, yb)
    if ya.numel():
        go = torch.randn(ya.shape, generator=g).to(D0)
        ya.backward(go); yb.backward(go)
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-5)     # <= 9 taps of magnitude ~10 summed in another order


@pytest.mark.parametrize("shape,p", [((1, 78899, 96), 0.1), ((3, 1001, 7), 0.5), ((5,), 0.25)])
def test_dropout_add_hip_statistics_and_gradient(hip, shape, p):
    """selfocc_dropout_add_fwd / _bwd (the `self.dropout(output) + identity` tail of the attention / FFN blocks): every
    element of y is identity or identity + x / (1 - p); the kept fraction is 1 - p; the backward applies the SAME mask;
    the seed follows torch.manual_seed; eval mode / p = 0 are the plain sum."""
    from selfocc_amd.dropout import dropout_add
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(*shape, generator=g) + 3.0).to(D0).requires_grad_(True)        # no zeros: kept <=> y != identity
    idt = torch.randn(*shape, generator=g).to(D0).requires_grad_(True)
    torch.manual_seed(5)
    y = dropout_add(x, idt, p, True)
    kept = (y.detach() - idt.detach()).abs() > 0
    assert torch.allclose(y.detach()[kept], (idt + x / (1 - p)).detach()[kept], rtol=1e-6, atol=1e-6)
    assert torch.equal(y.detach()[~kept], idt.detach()[~kept])
    n = x.numel()
    if n > 1000:
        assert abs(kept.float().mean().item() - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-3
        # no structure along rows / columns: per-column keep rates are all close to 1 - p
        if x.dim() == 3:
            assert (kept.float().mean((0, 1)) - (1 - p)).abs().max() < 6 * (p * (1 - p) / (n / x.shape[-1])) ** 0.5 + 1e-3
    w = torch.randn(*shape, generator=g).to(D0)
    (y * w).sum().backward()
    assert torch.allclose(idt.grad, w)
    assert torch.allclose(x.grad, torch.where(kept, w / (1 - p), torch.zeros_like(w)), rtol=1e-6, atol=1e-7)
    torch.manual_seed(5)
    assert torch.equal(dropout_add(x, idt, p, True).detach(), y.detach())            # same seed, same mask
    assert not torch.equal(dropout_add(x, idt, p, True).detach(), y.detach()) or n < 8
    assert torch.equal(dropout_add(x, idt, p, False).detach(), (x + idt).detach())
    assert torch.equal(dropout_add(x, idt, 0.0, True).detach(), (x + idt).detach())
