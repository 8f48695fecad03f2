"""GPU: ONE training step of the head + losses against the reference's own step (tests/golden/train_step.npz, written by
make_golden.py::golden_train_step): the REAL model/head/neus_head/neus_head.py forward at the shipped nuscenes_occ head
configuration -> the shipped `loss_input_convertion` -> the REAL loss/multi_loss.py over the REAL loss classes wired by the
shipped `loss` list (config/nuscenes/nuscenes_occ.py:111-186) -> backward (train.py:219-239), with the absent sdfstudio fork
served by the declared restatement in differentiable float64 form.

Pinned here: every loss term, the total, and the gradients w.r.t. the three TPV planes, every field-MLP parameter,
`variance` and the dense field volume — i.e. the composition NeuSHead -> MultiLoss -> backward through selfocc_reproj_bwd /
ssim_bwd / eikonal / second-diff -> selfocc_render_bwd (atomic AND binned scatter) -> selfocc_field_volume_bwd (fused) or
the torch route.  The measured worst cases go to gpurun_out/train_step_parity.jsonl; the asserted bounds are ~10 x those.
"""
import copy
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from test_golden_head_gpu import Replay

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
D0 = torch.device("cuda:0")
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "train_step_parity.jsonl")

# measured on MI355X (profiles/r5_a_train_step_parity.jsonl; three routes): rendered maps <= 4.6e-6 of scale, loss terms <= 3.3e-7
# relative, gradients <= 4.8e-6 (max) / 6.5e-6 (rel-L2) of each tensor's scale -- the bounds are ~10 x those
FWD_TOL = 5e-5            # ms_depths / ms_colors / ms_accs / sem: max |a - b| / max |b|
LOSS_RTOL = 5e-6          # every loss term and the total, relative
GRAD_MAX_TOL = 5e-5       # max |g - g_ref| / max |g_ref| per tensor
GRAD_L2_TOL = 7e-5        # ||g - g_ref|| / ||g_ref|| per tensor


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_inputs", os.path.join(G, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_INPUTS_ALL = {}


def _inputs(z, spec, stem="train_step"):
    """the seeded images / label map of the generator, regenerated once per session and fixture and checked against its digests"""
    _INPUTS = _INPUTS_ALL.setdefault(stem, {})
    if not _INPUTS:
        imgs, sem, prev, nxt = _gen().train_step_inputs(spec)
        for k, v in imgs.items():
            dg = [float(v.double().sum()), float(v[0, :, :, ::97, ::101].double().sum())]
            assert np.allclose(dg, z[f'digest.{k}'], rtol=1e-12), f"regenerated {k} differs from the generator's (torch RNG changed?)"
        assert [int(sem.sum()), int(sem[:, ::97, ::101].sum())] == z['digest.sem'].tolist()
        assert np.array_equal(prev, z['img2prevImg']) and np.array_equal(nxt, z['img2nextImg'])
        _INPUTS.update(imgs={k: v.to(D0) for k, v in imgs.items()}, sem=sem.to(D0))
    return _INPUTS['imgs'], _INPUTS['sem']


def _log(**m):
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps({k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in m.items()}) + "\n")
    except OSError:
        pass


def _errs(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    assert tuple(a.shape) == tuple(b.shape), (tuple(a.shape), tuple(b.shape))
    sc = b.abs().max().item()
    return (a - b).abs().max().item() / max(sc, 1e-30), ((a - b).norm() / b.norm().clamp_min(1e-30)).item(), sc


@pytest.mark.parametrize("scatter,fused_field", [('atomic', True), ('binned', True), ('binned', False)])
def test_train_step_losses_and_gradients_vs_reference(hip, monkeypatch, scatter, fused_field):
    _run(monkeypatch, "train_step", scatter, fused_field)


@pytest.mark.parametrize("scatter,fused_field", [('atomic', True), ('binned', True), ('binned', False)])
@pytest.mark.parametrize("stem", ["train_step_occ_bev", "train_step_kitti_raw"])
def test_train_step_of_the_other_shipped_compositions_vs_reference(hip, monkeypatch, stem, scatter, fused_field):
    """round 6: config/nuscenes/nuscenes_occ_bev.py (BEV field, SemLossMS + SoftSparsityLoss on the shifted uniform-SDF
    lattice) and config/kitti_raw/kitti_raw_depth.py (one camera, color_dims = 0, ReprojLossMonoMultiNew + EdgeLoss3DMS) —
    head and loss dictionaries read from the shipped config files by make_golden.py::golden_train_step_variants, held to the
    bounds of the nuscenes_occ step."""
    _run(monkeypatch, stem, scatter, fused_field)


def _run(monkeypatch, stem, scatter, fused_field):
    from selfocc_amd.registry import MODELS, OPENOCC_LOSS
    import selfocc_amd.model, selfocc_amd.loss  # noqa: F401
    z = np.load(os.path.join(G, stem + ".npz"))
    cfg = json.load(open(os.path.join(G, stem + "_cfg.json")))
    spec = cfg['spec']
    tpv = cfg['head'].get('tpv', True)
    head = MODELS.build(dict(type='NeuSHead', **copy.deepcopy(cfg['head'])))
    sd = {k[3:].replace('model.field.net.density_net', 'model.field.density_net'): torch.tensor(z[k])
          for k in z.files if k.startswith('sd.')}
    head.load_state_dict(sd, strict=True)
    head = head.to(D0).train()
    head.model.field.fused_volume = fused_field
    rep = [torch.tensor(z[f'rep{i}']).to(D0).requires_grad_(True) for i in range(3 if tpv else 1)]
    imgs, sem = _inputs(z, spec, stem)
    metas = [dict(img2lidar=list(z['img2lidar']), temImg2lidar=list(z['temImg2lidar']), img2prevImg=z['img2prevImg'],
                  img2nextImg=z['img2nextImg'], sem=sem)]
    loss_func = OPENOCC_LOSS.build(copy.deepcopy(cfg['loss']))          # the shipped `loss` dict, through our registry
    monkeypatch.setenv('SELFOCC_RB_SCATTER', scatter)
    os.environ['eval'] = 'false'
    draws = [z['draw.t_rand'], z['draw.bkgd']] + ([z['draw.shift']] if 'draw.shift' in z.files else [])
    rp = Replay(draws)
    orig_rand, orig_rand_like = torch.rand, torch.rand_like
    monkeypatch.setattr(torch, 'rand', rp.rand)
    monkeypatch.setattr(torch, 'rand_like', rp.rand_like)
    np.random.seed(spec['seed_np'])
    # ---- train.py:219-239 ----
    result_dict = head(rep if tpv else rep[0], metas, global_iter=spec['global_iter'])
    monkeypatch.setattr(torch, 'rand', orig_rand)
    monkeypatch.setattr(torch, 'rand_like', orig_rand_like)
    assert rp.used == len(draws)
    vol = head.model.field.volume
    vol.sdf.retain_grad()
    if vol.feat is not None:
        vol.feat.retain_grad()
    loss_input = {'curr_imgs': imgs['curr_imgs'], 'prev_imgs': imgs['prev_imgs'], 'next_imgs': imgs['next_imgs'],
                  'curr_feats': imgs['curr_imgs'], 'prev_feats': imgs['prev_imgs'], 'next_feats': imgs['next_imgs'],
                  'metas': metas, 'color_imgs': imgs['color_imgs']}
    for k, v in cfg['loss_input_convertion'].items():
        loss_input[k] = result_dict[v]
    loss, loss_dict = loss_func(loss_input)
    loss.backward()
    torch.cuda.synchronize()

    tag = f'{stem}:{scatter}/{"fused" if fused_field else "torch"}-field'
    # forward: the rendered maps the losses consume
    fwd = [(k, result_dict[k][0], z[f'out.{k}.0']) for k in ('ms_depths', 'ms_colors', 'ms_accs', 'sem')
           if f'out.{k}.0' in z.files and z[f'out.{k}.0'].size]
    if 'out.uniform_sdf' in z.files:
        fwd.append(('uniform_sdf', result_dict['uniform_sdf'], z['out.uniform_sdf']))
    for k, got, ref in fwd:
        emax, el2, sc = _errs(got, ref)
        _log(where=tag, kind='forward', key=k, err_max=emax, err_l2=el2, scale=sc)
        assert emax <= FWD_TOL, (tag, k, emax)
    # every loss term and the total
    terms = {k[5:]: float(z[k]) for k in z.files if k.startswith('loss.') and k != 'loss.total'}
    assert set(loss_dict) == set(terms), (sorted(loss_dict), sorted(terms))
    for k, ref in dict(terms, total=float(z['loss.total'])).items():
        got = float(loss.detach()) if k == 'total' else float(loss_dict[k])
        e = abs(got - ref) / abs(ref)
        _log(where=tag, kind='loss', key=k, err_rel=e, ref=ref, got=got)
        assert e <= LOSS_RTOL, (tag, k, got, ref)
    # gradients: planes, field MLP, variance, the dense volume
    g_vol = vol.sdf.grad[None] if vol.feat is None else torch.cat([vol.sdf.grad[None], vol.feat.grad.permute(3, 0, 1, 2)], 0)   # (1 + color_dims, H, W, D)
    pairs = [(f'rep{i}', rep[i].grad, z[f'grad.rep{i}']) for i in range(len(rep))] + [('volume', g_vol, z['grad.volume'])]
    floors = {}
    for n, p in head.named_parameters():
        rn = n.replace('model.field.density_net', 'model.field.net.density_net')
        pairs.append((n, p.grad, z['grad.sd.' + rn]))
        if 'gradscale.sd.' + rn in z.files:       # a scalar that sums signed terms: compared on the scale of what was summed
            floors[n] = float(z['gradscale.sd.' + rn])
    worst = {}
    for name, g, ref in pairs:
        assert g is not None and torch.isfinite(g).all(), (tag, name)
        emax, el2, sc = _errs(g, ref)
        if floors.get(name, 0.0) > sc:
            emax, el2, sc = emax * sc / floors[name], el2 * sc / floors[name], floors[name]
        _log(where=tag, kind='grad', key=name, err_max=emax, err_l2=el2, scale=sc)
        worst[name] = (emax, el2)
    bad = {n: e for n, e in worst.items() if e[0] > GRAD_MAX_TOL or e[1] > GRAD_L2_TOL}
    assert not bad, (tag, bad)
