"""CPU: bricks.stacked_view — the parameters of several Linears aliased into ONE stacked weight (no torch.cat per call) stay
ordinary parameters: names, values, gradients, optimiser updates, load_state_dict, deepcopy, and re-aliasing after `.to()`."""
import copy

import torch
import torch.nn as nn

from selfocc_amd.model.bricks import stacked_view


def _lins():
    torch.manual_seed(0)
    return nn.Linear(8, 6), nn.Linear(8, 3)


def test_values_names_and_aliasing():
    a, b = _lins()
    wa, wb = a.weight.detach().clone(), b.weight.detach().clone()
    w = stacked_view([a.weight, b.weight])
    assert torch.equal(w, torch.cat([wa, wb], 0)) and torch.equal(a.weight, wa) and torch.equal(b.weight, wb)
    assert w.data_ptr() == a.weight.data_ptr() and b.weight.data_ptr() == a.weight.data_ptr() + a.weight.numel() * 4
    w2 = stacked_view([a.weight, b.weight])                   # second call: already aliased, no new buffer
    assert w2.data_ptr() == w.data_ptr()
    assert set(dict(a.named_parameters())) == {'weight', 'bias'}
    # 1-D parameters (biases)
    bb = stacked_view([a.bias, b.bias])
    assert torch.equal(bb, torch.cat([a.bias.detach(), b.bias.detach()])) and bb.data_ptr() == a.bias.data_ptr()


def test_in_place_updates_are_seen_and_gradients_stay_separate():
    a, b = _lins()
    w = stacked_view([a.weight, b.weight])
    x = torch.randn(5, 8)
    opt = torch.optim.AdamW(list(a.parameters()) + list(b.parameters()), lr=0.1)
    (a(x).sum() + 2 * b(x).sum()).backward()
    assert a.weight.grad.shape == (6, 8) and b.weight.grad.shape == (3, 8)
    assert a.weight.grad.untyped_storage().data_ptr() != b.weight.grad.untyped_storage().data_ptr()
    before = w.clone()
    opt.step()
    w_after = stacked_view([a.weight, b.weight])
    assert w_after.data_ptr() == w.data_ptr() and not torch.equal(w_after, before)
    assert torch.equal(w_after, torch.cat([a.weight.detach(), b.weight.detach()], 0))
    # load_state_dict copies in place: the aliasing survives
    sd = {k: torch.randn_like(v) for k, v in a.state_dict().items()}
    a.load_state_dict(sd)
    assert torch.equal(stacked_view([a.weight, b.weight])[:6], sd['weight']) and stacked_view([a.weight, b.weight]).data_ptr() == w.data_ptr()


def test_moved_or_replaced_parameters_are_realiased_and_non_parameters_refused():
    a, b = _lins()
    w = stacked_view([a.weight, b.weight])
    a = a.double().float()                                    # `.data =` new tensors: aliasing broken
    w2 = stacked_view([a.weight, b.weight])
    assert w2.data_ptr() != w.data_ptr() and torch.equal(w2, torch.cat([a.weight.detach(), b.weight.detach()], 0))
    assert b.weight.data_ptr() == a.weight.data_ptr() + a.weight.numel() * 4
    c = copy.deepcopy(nn.ModuleList([a, b]))                  # deepcopy keeps the sharing inside the copy
    w3 = stacked_view([c[0].weight, c[1].weight])
    assert w3.data_ptr() == c[0].weight.data_ptr() != w2.data_ptr()
    assert stacked_view([a.weight * 1.0, b.weight]) is None   # not a leaf parameter (functional_call views): the caller cats
    assert stacked_view([a.weight, nn.Parameter(torch.zeros(3, 7))]) is None
