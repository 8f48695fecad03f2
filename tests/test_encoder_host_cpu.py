"""CPU: host logic of the encoder layers that needs no GPU — the plane bookkeeping of TPVFormerLayer (round 3), the torch
formula of the feature flatten and its hand-written backward (the CUDA path of the same Function is tested on the GPU)."""
import torch

from selfocc_amd.model.encoder import tpvformer as T


def test_planes_keep_their_concatenated_source():
    t = torch.randn(1, 12, 4, requires_grad=True)
    sizes = [5, 4, 3]
    p = T._as_planes(t, sizes)
    assert isinstance(p, tuple) and [x.shape[1] for x in p] == sizes
    assert T._as_cat(p) is t                              # no copy forward, no cat / split pair in the autograd graph
    assert T._as_planes(p, sizes) is p and T._as_cat(t) is t
    loose = [torch.randn(1, n, 4) for n in sizes]         # planes that are not views of one tensor: a real cat
    assert torch.equal(T._as_cat(loose), torch.cat(loose, 1))
    # gradients through (planes of t) == gradients through t
    w = torch.randn(1, 12, 4)
    (g1,) = torch.autograd.grad((T._as_cat(T._as_planes(t, sizes)) * w).sum(), t)
    (g2,) = torch.autograd.grad((torch.cat(torch.split(t, sizes, 1), 1) * w).sum(), t)
    assert torch.equal(g1, g2)


def test_flatten_feats_function_backward_matches_autograd_cpu():
    g = torch.Generator().manual_seed(0)
    shapes = [(6, 5), (3, 2)]
    cams = torch.randn(2, 8, generator=g, requires_grad=True)
    lvls = torch.randn(3, 8, generator=g, requires_grad=True)
    fa = [torch.randn(2, 2, 8, h, w, generator=g).requires_grad_(True) for h, w in shapes]
    fb = [f.detach().clone().requires_grad_(True) for f in fa]
    ya = T._FlattenFeats.apply(cams, lvls, *fa)
    yb = T._flatten_feats_torch(cams, lvls, fb)
    assert torch.equal(ya, yb) and ya.shape == (2, 36, 2, 8)
    go = torch.randn(ya.shape, generator=g)
    ga = torch.autograd.grad(ya, [cams, lvls] + fa, go)
    gb = torch.autograd.grad(yb, [cams, lvls] + fb, go)
    for a, b in zip(ga, gb):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    assert torch.all(ga[1][2] == 0)                      # a level embedding beyond the maps handed in


def test_colsum_two_stage():
    t = torch.randn(3, 1000, 5)
    assert torch.allclose(T._colsum(t), t.sum(1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(T._colsum(t[:, :7]), t[:, :7].sum(1), rtol=1e-5, atol=1e-5)
