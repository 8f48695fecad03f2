"""GPU: K = 4 training iterations with gradient accumulation against the reference's own trajectory
(tests/golden/train_steps_k.npz, written by make_golden.py::golden_train_steps): the REAL neus_head.py + REAL MultiLoss driven
exactly as train.py:219-254 — ``loss / grad_accumulation``, ``backward()``, every second iteration
``clip_grad_norm_(grad_max_norm)`` + ``AdamW.step()`` + ``zero_grad()`` — with a fresh cellular lattice (numpy RNG), jitter and
random background (torch RNG) per iteration and ``global_iter`` advancing.  Two trajectories: the shipped optimizer dict
(config/_base_/optimizer.py) and the same with lr x 100, where a stale lattice / inv_s / workspace / un-zeroed gradient shows
in the NEXT iteration's losses (at the shipped 2e-5 two steps move the losses by about the tolerance).

What a single step (test_golden_train_step_gpu.py) cannot catch and this does: state carried from one iteration to the next —
cached lattices, ``variance`` -> ``inv_s`` on the device, the render-backward brick / scatter workspaces, gradient
accumulation into existing ``.grad``s, the optimiser seeing the accumulated gradient.
Asserted: the lattice of every iteration (bit-exact), every loss term of every iteration, the clipped gradient norm of every
optimiser step, AdamW's first-moment state of every parameter (linear in the gradients: no amplification), and the final
parameters (on the elements whose gradient is above the noise floor: Adam moves a noise-level gradient by +-lr whatever its
sign — those are bounded by the step size instead).  Measured values go to gpurun_out/train_steps_parity.jsonl."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from test_golden_head_gpu import Replay
from test_golden_train_step_gpu import _inputs, _errs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
D0 = torch.device("cuda:0")
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "train_steps_parity.jsonl")

LOSS_RTOL = 1e-5           # every loss term of every iteration, relative
NORM_RTOL = 1e-5           # the clipped gradient norm of every optimiser step
GRAD_MAX_TOL, GRAD_L2_TOL = 5e-5, 7e-5      # the accumulated gradient at every optimiser step (the bounds of the one-step test)
STATE_TOL = 5e-5           # AdamW exp_avg: max |m - m_ref| / max |m_ref| per tensor
PARAM_TOL = 5e-5           # final parameters on reliable elements: max |p - p_ref| / max(|p_ref|, update scale)


def _log(**m):
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps({k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in m.items()}) + "\n")
    except OSError:
        pass


def _ours(name):
    return name.replace('model.field.net.density_net', 'model.field.density_net')


@pytest.mark.parametrize("scatter,fused_field", [('binned', True), ('atomic', True), ('binned', False)])
@pytest.mark.parametrize("traj", [0, 1])
def test_k_training_iterations_with_accumulation_vs_reference(hip, monkeypatch, traj, scatter, fused_field):
    from selfocc_amd.registry import MODELS, OPENOCC_LOSS
    import selfocc_amd.model, selfocc_amd.loss  # noqa: F401
    z = np.load(os.path.join(G, "train_steps_k.npz"))
    cfg = json.load(open(os.path.join(G, "train_steps_k_cfg.json")))
    spec, ks = cfg['spec'], cfg['steps']
    pre, lr_mult = f't{traj}', ks['lr_mults'][traj]
    head = MODELS.build(dict(type='NeuSHead', **copy.deepcopy(cfg['head'])))
    sd = {_ours(k[8:]): torch.tensor(z[k]) for k in z.files if k.startswith('init.sd.')}
    head.load_state_dict(sd, strict=True)
    head = head.to(D0).train()
    head.model.field.fused_volume = fused_field
    rep = [torch.nn.Parameter(torch.tensor(z[f'init.rep{i}']).to(D0)) for i in range(3)]
    named = [(f'rep{i}', r) for i, r in enumerate(rep)] + [('sd.' + n, p) for n, p in head.named_parameters()]
    params = [p for _, p in named]
    opt_cfg = dict(cfg['optimizer'])
    assert opt_cfg.pop('type') == 'AdamW'
    optimizer = torch.optim.AdamW(params, **dict(opt_cfg, lr=opt_cfg['lr'] * lr_mult))
    imgs, sem = _inputs(z, spec)                       # the images of train_step.npz: shared cache
    metas = [dict(img2lidar=list(z['img2lidar']), temImg2lidar=list(z['temImg2lidar']), img2prevImg=z['img2prevImg'],
                  img2nextImg=z['img2nextImg'], sem=sem)]
    loss_func = OPENOCC_LOSS.build(copy.deepcopy(cfg['loss']))
    monkeypatch.setenv('SELFOCC_RB_SCATTER', scatter)
    os.environ['eval'] = 'false'
    np.random.seed(ks['seed_np'])
    orig_rand = torch.rand
    tag = f'{pre}(lr x{lr_mult:g})/{scatter}/{"fused" if fused_field else "torch"}-field'
    n_steps, worst_loss = 0, 0.0
    for it in range(ks['K']):
        global_iter = ks['first_iter'] + it
        rp = Replay([z[f'{pre}.it{it}.draw.t_rand'], z[f'{pre}.it{it}.draw.bkgd']])
        monkeypatch.setattr(torch, 'rand', rp.rand)
        # ---- train.py:219-254 ----
        result_dict = head(rep, metas, global_iter=global_iter)
        monkeypatch.setattr(torch, 'rand', orig_rand)
        assert rp.used == 2
        assert np.array_equal(result_dict['ms_rays'].cpu().numpy(), z[f'{pre}.it{it}.ms_rays']), (tag, it, "the lattice of this iteration")
        loss_input = {'curr_imgs': imgs['curr_imgs'], 'prev_imgs': imgs['prev_imgs'], 'next_imgs': imgs['next_imgs'],
                      'curr_feats': imgs['curr_imgs'], 'prev_feats': imgs['prev_imgs'], 'next_feats': imgs['next_imgs'],
                      'metas': metas, 'color_imgs': imgs['color_imgs']}
        for k, v in cfg['loss_input_convertion'].items():
            loss_input[k] = result_dict[v]
        loss, loss_dict = loss_func(loss_input)
        loss = loss / ks['grad_accumulation']
        loss.backward()
        terms = {k[len(f'{pre}.it{it}.loss.'):]: float(z[k]) for k in z.files if k.startswith(f'{pre}.it{it}.loss.')}
        assert set(loss_dict) | {'total'} == set(terms)
        for k, ref in terms.items():
            got = float(loss.detach()) if k == 'total' else float(loss_dict[k])
            e = abs(got - ref) / abs(ref)
            worst_loss = max(worst_loss, e)
            _log(where=tag, kind='loss', it=it, key=k, err_rel=e, ref=ref, got=got)
            assert e <= LOSS_RTOL, (tag, it, k, got, ref)
        inv_s = float(head.model.field.inv_s())
        assert abs(inv_s - float(z[f'{pre}.it{it}.inv_s'])) <= 1e-5 * inv_s, (tag, it, inv_s)
        if (global_iter + 1) % ks['grad_accumulation'] == 0:
            gn = float(torch.nn.utils.clip_grad_norm_(params, cfg['grad_max_norm']))
            ref = float(z[f'{pre}.step{n_steps}.grad_norm'])
            _log(where=tag, kind='grad_norm', step=n_steps, err_rel=abs(gn - ref) / ref, ref=ref, got=gn)
            assert abs(gn - ref) <= NORM_RTOL * ref, (tag, n_steps, gn, ref)
            for n, p in named:       # the accumulated gradient the optimiser is about to see (planes: every 8th row)
                rn = n.replace('model.field.density_net', 'model.field.net.density_net')
                g_max, g_l2, g_sc = _errs(p.grad[0, ::8] if n.startswith('rep') else p.grad, z[f'{pre}.step{n_steps}.grad.{rn}'])
                _log(where=tag, kind='step_grad', step=n_steps, key=n, err_max=g_max, err_l2=g_l2, scale=g_sc)
                if os.environ.get('SO_TS_NOASSERT') != '1':
                    assert g_max <= GRAD_MAX_TOL and g_l2 <= GRAD_L2_TOL, (tag, n_steps, n, g_max, g_l2)
            optimizer.step()
            optimizer.zero_grad()
            n_steps += 1
    assert n_steps == ks['K'] // ks['grad_accumulation']
    torch.cuda.synchronize()
    lr = opt_cfg['lr'] * lr_mult
    bad = {}
    for n, p in named:
        rn = n.replace('model.field.density_net', 'model.field.net.density_net')        # the reference's state-dict key
        m_max, m_l2, m_sc = _errs(optimizer.state[p]['exp_avg'], z[f'{pre}.exp_avg.{rn}'])
        ref, init = torch.tensor(z[f'{pre}.final.{rn}']).double(), torch.tensor(z[f'init.{rn}']).double()
        got = p.detach().double().cpu()
        rel = torch.tensor(np.unpackbits(z[f'{pre}.reliable.{rn}'])[:ref.numel()].astype(bool)).reshape(ref.shape)
        scale = max(float(ref.abs().max()), 1e-30)
        e_rel = float((got - ref)[rel].abs().max()) / scale if rel.any() else 0.0
        e_unrel = float((got - ref)[~rel].abs().max()) if (~rel).any() else 0.0
        # the update itself (what the optimiser did), relative to its own size, on the reliable elements
        upd_ref = (ref - init)[rel]
        e_upd = float(((got - init)[rel] - upd_ref).abs().max()) / max(float(upd_ref.abs().max()), 1e-30) if rel.any() else 0.0
        _log(where=tag, kind='final', key=n, exp_avg_err_max=m_max, exp_avg_err_l2=m_l2, param_err_reliable=e_rel,
             update_err_reliable=e_upd, param_abs_err_unreliable=e_unrel, reliable_frac=float(rel.float().mean()), lr=lr)
        if m_max > STATE_TOL or e_rel > PARAM_TOL or e_unrel > 2.2 * lr * n_steps or e_upd > 0.05:
            bad[n] = (m_max, e_rel, e_upd, e_unrel)
    assert not bad, (tag, bad)
    _log(where=tag, kind='summary', worst_loss_err_rel=worst_loss)
