"""CPU: pin the C oracle of MSDA against the torch formulation the reference itself uses
on CPU (multi_scale_deformable_attn_pytorch -> F.grid_sample) incl. autograd gradients."""
import pytest
import torch

import oracle
from oracle import torch_port as tp


def make_case(bs, nq, heads, d, shapes, P, seed=0, spread=0.1):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor(shapes, dtype=torch.int64)
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    L = shapes.shape[0]
    value = torch.randn(bs, nv, heads, d, generator=g)
    loc = torch.rand(bs, nq, heads, L, P, 2, generator=g) * (1 + 2 * spread) - spread  # ~17 % outside
    attw = torch.softmax(torch.randn(bs, nq, heads, L * P, generator=g), -1).reshape(bs, nq, heads, L, P)
    return value, shapes, starts, loc, attw


CASES = [
    (1, 50, 6, 16, [[12, 25], [6, 13], [3, 7], [2, 4]], 8),     # cross-attn like (4 levels)
    (2, 33, 6, 16, [[9, 9], [3, 9], [9, 3]], 12),                 # cross-view self-attn like
    (1, 7, 2, 8, [[5, 4]], 3),
    (1, 5, 1, 32, [[4, 6], [2, 3]], 1),
]


@pytest.mark.parametrize("case", CASES)
def test_msda_fwd_oracle_vs_grid_sample(case):
    value, shapes, starts, loc, attw = make_case(*case)
    ref = tp.msda_port(value, shapes, loc, attw)
    got = oracle.msda_fwd(value, shapes, starts, loc, attw)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", CASES[:3])
def test_msda_bwd_oracle_vs_autograd(case):
    value, shapes, starts, loc, attw = make_case(*case, seed=1)
    value.requires_grad_(True); loc.requires_grad_(True); attw.requires_grad_(True)
    out = tp.msda_port(value, shapes, loc, attw)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
    out.backward(g)
    gv, gl, ga = oracle.msda_bwd(value.detach(), shapes, starts, loc.detach(), attw.detach(), g)
    assert torch.allclose(gv, value.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(ga, attw.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gl, loc.grad, rtol=1e-3, atol=1e-4)


def test_msda_edge_locations():
    """exact borders, far outside, pixel centres"""
    value, shapes, starts, _, _ = make_case(1, 1, 1, 4, [[4, 5]], 1)
    pts = torch.tensor([[0.0, 0.0], [1.0, 1.0], [-0.5, 0.3], [1.5, 0.2], [0.1, 0.125], [0.9999, 0.5],
                        [-0.1, -0.1], [0.5 / 5, 0.5 / 4]])
    loc = pts.reshape(1, 8, 1, 1, 1, 2)
    attw = torch.ones(1, 8, 1, 1, 1)
    ref = tp.msda_port(value, shapes, loc, attw)
    got = oracle.msda_fwd(value, shapes, starts, loc, attw)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6)
