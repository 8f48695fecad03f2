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
    stem = "head_occ" if tag == 'occ' else "head"      # 'occ' = the shipped nuscenes_occ head (its own file)
    z = np.load(os.path.join(G, stem + ".npz"))
    cfg = json.load(open(os.path.join(G, stem + "_cfg.json")))[tag]
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
RENDER = ('ms_depths', 'ms_colors', 'ms_accs', 'sem', 'vis_normal')                                 # composited: 1e-4 relative

LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "head_parity.jsonl")


def _log(where, k, i, **m):
    """measured worst case per (call, key) -> gpurun_out/head_parity.jsonl (the asserted bounds below are these x 10)"""
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps(dict(where=where, key=k, idx=i, **{a: (float(b) if b is not None else None) for a, b in m.items()})) + "\n")
    except OSError:
        pass


def rel_err(a, b, floor):
    """max over elements of |a - b| / max(|b|, floor): relative where the reference is large, absolute (in units of
    `floor`) where it is small"""
    return ((a - b).abs() / b.abs().clamp_min(floor)).max().item()


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
            sc = max(1e-30, b.abs().max().item())
            if k == 'ray_indices':
                assert a.dtype == b.dtype == torch.int64 and torch.equal(a, b), tag
            elif k in GEOM:
                e = rel_err(a, b, sc)
                _log(where, k, i, err=e, scale=sc)
                assert close(a, b, 1e-5, 1e-5).all(), (tag, (a - b).abs().max().item())
            elif k in FIELD:
                e = rel_err(a, b, sc)                       # measured <= 1.3e-6 of the tensor's scale
                _log(where, k, i, err=e, scale=sc)
                assert e <= 2e-5 and close(a, b, 1e-4, 2e-5).all(), (tag, e, (a - b).abs().max().item())
            elif k == 'sem' and a.dtype == torch.int64:
                pass        # arg-max of the logits: checked against the logits by the caller
            elif k == 'ms_max_depths':
                # arg-max over w / delta.  An (almost) tie could in principle resolve to the neighbouring sample; on the recorded
                # calls it never does — every ray of every call agrees — and that is what is asserted (round-5 review: "assert
                # what is measured"; the bound was 99 %)
                frac = close(a, b, 1e-5, 1e-5).float().mean().item()
                _log(where, k, i, frac=frac)
                assert frac == 1.0, (tag, frac)
            elif k == 'eik_grad':
                # trilinear gradients jump across voxel faces: a sample within rounding of a face may sit next door
                ok = close(a, b, 1e-4, 1e-4 * b.abs().max().item()).all(-1)
                e = rel_err(a[ok], b[ok], sc) if ok.any() else 0.0        # measured: every sample, <= 1e-6 of the scale
                _log(where, k, i, frac=ok.float().mean().item(), err=e)
                assert ok.float().mean() >= 0.999 and e <= 1e-5, (tag, ok.float().mean().item(), e)
            elif k == 'weights':
                # per-SAMPLE weights w_i = alpha_i T_i: alpha subtracts two sigmoids that agree to ~1e-5 in free space, so a
                # float32 evaluation (the reference's as much as ours) carries ~1e-7 of ABSOLUTE rounding noise per sample
                # whatever the weight's size (tests/util.py: parity_report); their sums (ms_accs), weighted sums (ms_depths,
                # ms_colors, sem) are held to 1e-4 relative below.  Measured: <= 1.5e-6 absolute on every element of every call.
                ea = (a - b).abs().max().item()
                _log(where, k, i, err=rel_err(a, b, 1e-2 * sc), err_abs=ea, scale=sc)
                assert ea <= 1.5e-5, (tag, ea)
            elif k in RENDER:
                # north_star: rendered depth / RGB within 1e-4 relative.  `err` = max |a - b| / max(|b|, scale_floor) with
                # scale_floor = 1e-2 of the tensor's largest value (a colour channel of 1e-9 carries no 1e-4-relative
                # information in float32); EVERY element of EVERY call is held to it (round 3: 99.5 % of them, 2e-3 of the
                # scale on the rest).  Measured worst case: 4.8e-5 (`sem` of the shipped nuscenes_occ head, 256 samples).
                e = rel_err(a, b, 1e-2 * sc)
                _log(where, k, i, err=e, err_abs=(a - b).abs().max().item(), scale=sc)
                assert e <= RENDER_TOL, (tag, e, (a - b).abs().max().item())
            else:
                raise AssertionError(f"no comparison rule for key {k!r}")


RENDER_TOL = 1e-4


def _sem_argmax_consistent(ours, ref):
    """forward_occ's `sem` = argmax(logits): identical wherever the reference's top-2 logits are > 1e-4 apart"""
    top2 = ref['logits'].topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert torch.equal(ours['sem'].cpu()[clear], ref['sem'][clear])
    assert ours['sem'].dtype == torch.int64


@pytest.mark.parametrize("tag", ['tpv', 'bev', 'occ'])
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


@pytest.mark.parametrize("tag,name,batch", [('tpv', 'render0', 0), ('tpv', 'render50', 50), ('bev', 'render0', 0),
                                            ('bev', 'render50', 50), ('occ', 'render0', 0)])
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


@pytest.mark.parametrize("tag", ['tpv', 'bev', 'occ'])
def test_head_forward_occ_vs_reference_head(hip, tag):
    z, cfg, head, rep, metas = _load(tag)
    head.eval()
    with torch.no_grad():
        for name, kw in (('occ', dict(aabb=[-6.0, -5.0, -0.5, 6.0, 7.0, 2.5], resolution=0.5)), ('occdef', {})):
            if tag == 'occ' and name == 'occdef':
                continue            # the shipped-shape fixture records the explicit-box call only
            out = head.forward_occ(rep, metas, **kw)
            assert out.pop('rep') is rep
            ref = _ref_dict(z, f'{tag}.{name}')
            compare(out, ref, f'{tag}.{name}')
            if 'sem' in ref:
                _sem_argmax_consistent(out, ref)
