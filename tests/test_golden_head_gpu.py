"""GPU: our NeuSHead against the dicts the REAL model/head/neus_head/neus_head.py returned on CPU
(tests/golden/head.npz, written by make_golden.py::golden_head with the sdfstudio fork stood in by
oracle/torch_port.py).  Pinned here: everything neus_head.py does itself — ray construction (:308-352, :473-530),
ts / deltas / max-depth post-math (:571-587, :430-438), get_uniform_sdf (:265-293), forward_occ (:237-263),
two-split and dict assembly (:459-471, :667-713).  The fork's NeuS internals stay "parity unpinned".
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
D0 = torch.device("cuda:0")


class Replay:
    """torch.rand / torch.rand_like stand-ins that return the reference run's recorded draws (by element count and
    last dimension), so that jitter, random backgrounds and the lattice shift are the same numbers on both sides."""

    def __init__(self, draws):
        self.draws = [torch.tensor(d) for d in draws]
        self.used = 0

    def _find(self, shape):
        n = int(np.prod(shape))
        for d in self.draws:
            if d.numel() == n and (len(shape) < 2 or d.shape[-1] == shape[-1]):
                self.used += 1
                return d.reshape(shape)
        raise AssertionError(f"no recorded draw of shape {tuple(shape)}")

    def rand(self, *size, device=None, **kw):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        return self._find(shape).to(device or 'cpu')

    def rand_like(self, t, **kw):
        return self._find(tuple(t.shape)).to(t.device)


def _load(tag, **over):
    from selfocc_amd.registry import MODELS
    import selfocc_amd.model  # noqa: F401
    z = np.load(os.path.join(G, "head.npz"))
    cfg = json.load(open(os.path.join(G, "head_cfg.json")))[tag]
    cfg = dict(copy.deepcopy(cfg), **over)
    head = MODELS.build(dict(type='NeuSHead', **cfg))
    sd = {k[len(tag) + 4:].replace('model.field.net.density_net', 'model.field.density_net'): torch.tensor(z[k])
          for k in z.files if k.startswith(f'{tag}.sd.')}
    head.load_state_dict(sd, strict=True)
    head = head.to(D0)
    reps = [torch.tensor(z[f'{tag}.rep{i}']).to(D0) for i in range(3 if cfg['tpv'] else 1)]
    rep = reps if cfg['tpv'] else reps[0]
    metas = [dict(img2lidar=list(z[f'{tag}.img2lidar']), temImg2lidar=list(z[f'{tag}.temImg2lidar']))]
    return z, cfg, head, rep, metas


def _ref_dict(z, prefix):
    """npz arrays of one recorded call -> {key: tensor | [tensors] | None}"""
    out = {}
    for k in z.files:
        if not k.startswith(prefix + '.') or k.startswith(prefix + '.draw.'):
            continue
        name = k[len(prefix) + 1:]
        parts = name.split('.')
        if len(parts) == 1:
            out[name] = torch.tensor(z[k])
        elif parts[1] == 'none':
            out[parts[0]] = None
        elif parts[1] == 'len':
            out.setdefault(parts[0], [None] * int(z[k]))
        else:
            out.setdefault(parts[0], [None] * int(z[f'{prefix}.{parts[0]}.len']))[int(parts[1])] = torch.tensor(z[k])
    return out


def _draws(z, prefix):
    return [z[k] for k in z.files if k.startswith(prefix + '.draw.')]


def close(a, b, rtol, atol):
    return (a - b).abs() <= atol + rtol * b.abs()


GEOM = ('ms_rays', 'origin', 'direction', 'direction_norm', 'ts', 'deltas', 'ms_fars', 'xyz')      # 1e-5: neus_head.py's own math
FIELD = ('uniform_sdf', 'sdf', 'logits', 'sample_sdf', 'second_grad')                              # volume MLP + trilinear lookup
RENDER = ('ms_depths', 'ms_colors', 'ms_accs', 'sem', 'weights', 'vis_normal')                      # composited (oracle tolerance)


def compare(ours, ref, where):
    assert set(ours) == set(ref), (where, sorted(set(ours) ^ set(ref)))
    for k, r in ref.items():
        o = ours[k]
        if r is None:
            assert o is None, (where, k)
            continue
        pairs = list(zip(o, r)) if isinstance(r, list) else [(o, r)]
        if isinstance(r, list):
            assert isinstance(o, (list, tuple)) and len(o) == len(r), (where, k)
        for i, (a, b) in enumerate(pairs):
            a = a.detach().cpu()
            assert tuple(a.shape) == tuple(b.shape), (where, k, i, tuple(a.shape), tuple(b.shape))
            tag = (where, k, i)
            if b.numel() == 0:          # colourless heads return (1, cams, rays, 0) `ms_colors`
                continue
            if k == 'ray_indices':
                assert a.dtype == b.dtype == torch.int64 and torch.equal(a, b), tag
            elif k in GEOM:
                assert close(a, b, 1e-5, 1e-5).all(), (tag, (a - b).abs().max().item())
            elif k in FIELD:
                assert close(a, b, 1e-4, 2e-5).all(), (tag, (a - b).abs().max().item())
            elif k == 'sem' and a.dtype == torch.int64:
                pass        # arg-max of the logits: checked against the logits by the caller
            elif k == 'ms_max_depths':
                # arg-max over w / delta: an (almost) tie may legitimately resolve to the neighbouring sample
                assert close(a, b, 1e-5, 1e-5).float().mean() >= 0.97, (tag, close(a, b, 1e-5, 1e-5).float().mean().item())
            elif k == 'eik_grad':
                # trilinear gradients jump across voxel faces: a sample within rounding of a face may sit next door
                ok = close(a, b, 1e-4, 1e-4 * b.abs().max().item()).all(-1)
                assert ok.float().mean() >= 0.999, (tag, ok.float().mean().item())
            elif k in RENDER:
                sc = max(1.0, b.abs().max().item())
                assert close(a, b, 1e-4, 2e-5 * sc).float().mean() >= 0.995, (tag, (a - b).abs().max().item())
                assert (a - b).abs().max().item() <= 2e-3 * sc, (tag, (a - b).abs().max().item())
            else:
                raise AssertionError(f"no comparison rule for key {k!r}")


def _sem_argmax_consistent(ours, ref):
    """forward_occ's `sem` = argmax(logits): identical wherever the reference's top-2 logits are > 1e-4 apart"""
    top2 = ref['logits'].topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert torch.equal(ours['sem'].cpu()[clear], ref['sem'][clear])
    assert ours['sem'].dtype == torch.int64


@pytest.mark.parametrize("tag", ['tpv', 'bev'])
def test_head_train_forward_vs_reference_head(hip, monkeypatch, tag):
    z, cfg, head, rep, metas = _load(tag)
    os.environ['eval'] = 'false'
    head.train()
    rp = Replay(_draws(z, f'{tag}.train'))
    monkeypatch.setattr(torch, 'rand', rp.rand)
    monkeypatch.setattr(torch, 'rand_like', rp.rand_like)
    np.random.seed(77)
    out = head(rep, metas, global_iter=7)
    monkeypatch.undo()
    assert rp.used == len(rp.draws)          # every recorded draw was consumed: same random inputs on both sides
    compare(out, _ref_dict(z, f'{tag}.train'), f'{tag}.train')


@pytest.mark.parametrize("tag", ['tpv', 'bev'])
def test_head_eval_forward_vs_reference_head(hip, monkeypatch, tag):
    z, cfg, head, rep, metas = _load(tag)
    os.environ['eval'] = 'true'
    try:
        head.eval()
        rp = Replay(_draws(z, f'{tag}.evalfwd'))
        monkeypatch.setattr(torch, 'rand', rp.rand)
        monkeypatch.setattr(torch, 'rand_like', rp.rand_like)
        with torch.no_grad():
            out = head(rep, metas)
        monkeypatch.undo()
        assert rp.used == len(rp.draws)
        compare(out, _ref_dict(z, f'{tag}.evalfwd'), f'{tag}.evalfwd')
    finally:
        os.environ['eval'] = 'false'


@pytest.mark.parametrize("tag", ['tpv', 'bev'])
@pytest.mark.parametrize("name,batch", [('render0', 0), ('render50', 50)])
def test_head_prepare_render_vs_reference_head(hip, monkeypatch, tag, name, batch):
    """eval_depth.py:165-166 — prepare() + render(batch): the reference's chunk loop and ours (one launch) agree"""
    z, cfg, head, rep, metas = _load(tag, render_normal=True)
    os.environ['eval'] = 'true'
    try:
        head.eval()
        rp = Replay(_draws(z, f'{tag}.{name}'))
        monkeypatch.setattr(torch, 'rand', rp.rand)
        with torch.no_grad():
            assert head.prepare(rep, metas) == {}
            out = head.render(metas, batch=batch)
        monkeypatch.undo()
        compare(out, _ref_dict(z, f'{tag}.{name}'), f'{tag}.{name}')
    finally:
        os.environ['eval'] = 'false'


@pytest.mark.parametrize("tag", ['tpv', 'bev'])
def test_head_forward_occ_vs_reference_head(hip, tag):
    z, cfg, head, rep, metas = _load(tag)
    head.eval()
    with torch.no_grad():
        for name, kw in (('occ', dict(aabb=[-6.0, -5.0, -0.5, 6.0, 7.0, 2.5], resolution=0.5)), ('occdef', {})):
            out = head.forward_occ(rep, metas, **kw)
            assert out.pop('rep') is rep
            ref = _ref_dict(z, f'{tag}.{name}')
            compare(out, ref, f'{tag}.{name}')
            if 'sem' in ref:
                _sem_argmax_consistent(out, ref)
