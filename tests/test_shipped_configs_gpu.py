"""GPU: ALL SEVEN experiment configs the reference ships (config/{nuscenes,kitti,kitti_raw}/*.py), each built through the
registries from its dumped JSON (scripts/shipped_cfg/*.json, written by scripts/dump_shipped_configs.py) at the SHIPPED
shapes — lifter, encoder (4 layers), NeuSHead, MultiLoss with the config's own loss list and `loss_input_convertion`:

  * one TRAINING iteration (train.py:219-239: forward, losses, backward) on a new frame under
    ``torch.cuda.set_sync_debug_mode("error")`` after one warm-up iteration; every loss term finite, a finite NON-ZERO
    gradient on every parameter of lifter / encoder / head;
  * then the config's EVALUATION entry with the reference's overrides (utils/config_tools.py:10-116 + the entry script the
    docs pair with it, docs/get_started.md:17-107): ``prepare`` + ``render`` for the depth / novel-depth configs,
    ``forward_occ`` + the Occ3D tail (eval_iou.py) or the SemanticKITTI tail (eval_iou_kitti.py) for the occupancy ones.

The pieces are pinned against the reference elsewhere (encoders, head, losses, one whole training step: test_golden_*);
this file pins the COMPOSITIONS the drop-in claim covers: round 5 executed three of the seven."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
D0 = torch.device("cuda:0")

NAMES = ['nuscenes_occ', 'nuscenes_occ_bev', 'nuscenes_depth', 'nuscenes_novel_depth', 'kitti_occ', 'kitti_novel_depth',
         'kitti_raw_depth']


class no_sync:
    def __enter__(self):
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")

    def __exit__(self, *a):
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()


def test_every_shipped_config_is_listed():
    import hotpath_common as hc
    have = sorted(f[:-5] for f in os.listdir(os.path.join(ROOT, "scripts", "shipped_cfg")) if f.endswith(".json"))
    assert have == sorted(NAMES) == sorted(hc.SHIPPED)


@pytest.mark.parametrize("name", NAMES)
def test_shipped_config_trains_and_evaluates(hip, name):
    import hotpath_common as hc
    torch.manual_seed(0)
    np.random.seed(0)
    # ---------------- training iteration: train.py:219-239 ----------------
    os.environ['eval'] = 'false'
    cfg = hc.shipped(name)
    mods = hc.build(cfg, D0, want_loss=True)
    lifter, encoder, head, loss_fn = mods
    for m in (lifter, encoder, head):
        m.train()
    # the shipped initialisation zeroes the `sampling_offsets` / `attention_weights` weights (mmcv's init_weights;
    # image_cross_attention.py:224-246), which makes d loss / d query — and with it the gradient of everything upstream of a
    # layer's queries through those two linears, e.g. the positional encoding — EXACTLY zero at iteration 0 in the reference
    # too.  A trained state is the interesting one: every all-zero parameter gets small seeded values.
    g = torch.Generator(device='cpu').manual_seed(3)
    with torch.no_grad():
        for m in (lifter, encoder, head):
            for p in m.parameters():
                if float(p.abs().max()) == 0.0:
                    p.copy_((0.02 * torch.randn(p.shape, generator=g)).to(p.device))
    params = [(f'{tag}.{n}', p) for tag, m in (('lifter', lifter), ('encoder', encoder), ('head', head))
              for n, p in m.named_parameters()]
    hc.train_iteration(mods, cfg, hc.frame_inputs(cfg, name, D0, seed=0), global_iter=0)         # warm-up: workspaces, constants
    for _, p in params:
        p.grad = None
    new = hc.frame_inputs(cfg, name, D0, seed=1)
    with no_sync():
        total, parts, out = hc.train_iteration(mods, cfg, new, global_iter=1)
    want_terms = [c['type'] for c in cfg['loss']['loss_cfgs']]
    assert len(parts) == len(want_terms), (sorted(parts), want_terms)
    assert torch.isfinite(total).all() and float(total) != 0.0
    for k, v in parts.items():
        assert np.isfinite(float(v)), (name, k, float(v))
    n_rays = cfg['num_rays'][0] * cfg['num_rays'][1]
    n_cams = cfg['model']['encoder']['num_cams']
    assert out['ms_depths'][0].shape == (1, n_cams, n_rays)
    assert out['weights'][0].numel() == n_rays * cfg['model']['head']['num_samples']
    bad = [n for n, p in params if p.requires_grad and (p.grad is None or not torch.isfinite(p.grad).all() or float(p.grad.abs().max()) == 0.0)]
    assert not bad, (name, bad[:8], len(bad))
    del mods, lifter, encoder, head, loss_fn, params, out, total, parts, new
    torch.cuda.empty_cache()

    # ---------------- evaluation entry with the reference's overrides ----------------
    os.environ['eval'] = 'true'
    try:
        ecfg = hc.shipped_for_eval(name)
        emods = hc.build(ecfg, D0)
        for m in emods[:3]:
            m.eval()
        state = {}
        kind = hc.SHIPPED[name]['eval']
        with torch.no_grad():
            hc.eval_entry(emods, ecfg, name, hc.frame_inputs(ecfg, name, D0, seed=2, want_images=False), state)     # warm-up
            new = hc.frame_inputs(ecfg, name, D0, seed=3, want_images=False)
            with no_sync():
                res = hc.eval_entry(emods, ecfg, name, new, state)
        nr = hc.NUM_RAYS[hc.SHIPPED[name]['dataset']]
        if kind in ('render', 'render_novel'):
            d = res['ms_depths'][0]
            assert d.shape == (1, n_cams, nr[0] * nr[1]) and torch.isfinite(d).all() and float(d.max()) > 0
            assert res['ms_max_depths'][0].shape == d.shape and torch.isfinite(res['ms_max_depths'][0]).all()
            assert torch.isfinite(res['ms_accs'][0]).all()
            if ecfg['model']['head']['color_dims'] >= 3:
                assert res['ms_colors'][0].shape == (1, n_cams, nr[0] * nr[1], 3) and torch.isfinite(res['ms_colors'][0]).all()
        elif kind == 'occ3d':
            assert res['sdf'].shape == (200, 200, 16) and torch.isfinite(res['sdf']).all()
            assert res['occ'].shape == (200, 200, 16) and res['sem_nus'].dtype == torch.int32
            assert int(res['sem_nus'].max()) <= 16 and int(res['sem_nus'].min()) >= 0
            miou, iou = state['miou']._after_epoch()
            assert np.isfinite(miou) and np.isfinite(iou)
        else:
            assert res['sdf'].shape == (256, 256, 32) and torch.isfinite(res['sdf']).all()
            assert int(res['occ'][..., 28:].sum()) == 0 and int(res['occ'][-6:].sum()) == 0
            _, iou = state['miou']._after_epoch()
            assert np.isfinite(iou)
    finally:
        os.environ['eval'] = 'false'


@pytest.mark.parametrize("name", NAMES)
def test_shipped_config_independent_routes_agree_at_the_shipped_size(hip, name, monkeypatch):
    """Parity at the FULL shipped size, where no reference run or oracle fits: the same training iteration (dropout off, same
    frame, same lattice / jitter / background draws) through two kernel routes that share only the lowest-level ops —
    (A) the defaults: merged offset | logit projections, camera-loop / fused MSDA with in-kernel prologue, value-gradient sink,
        fused tri-plane MLP, binned render-backward scatter;
    (B) the reference-shaped route: two Linears per attention, BEVCrossAttention's re-batch + the plain
        MultiScaleDeformableAttnFunction (mmcv's boundary) with torch softmax / locations, head-major gradients + copies,
        the op-by-op field MLP, the atomic scatter —
    must give the same loss terms and the same gradients.  The small fixtures pin both routes to the reference; this pins the
    routes to each other at 257 x 257 x 25..33 planes, 6 cameras / 28 800 rays (nuScenes) or 1 camera (KITTI), 256 samples."""
    import hotpath_common as hc
    from selfocc_amd.model import bricks
    from selfocc_amd.model.encoder.attention import BEVCrossAttention
    os.environ['eval'] = 'false'
    cfg = hc.shipped(name)

    def run(route_b):
        torch.manual_seed(0)
        np.random.seed(0)
        mods = hc.build(cfg, D0, want_loss=True)
        lifter, encoder, head, loss_fn = mods
        g = torch.Generator(device='cpu').manual_seed(3)
        with torch.no_grad():
            for m in (lifter, encoder, head):
                for p in m.parameters():
                    if float(p.abs().max()) == 0.0:
                        p.copy_((0.02 * torch.randn(p.shape, generator=g)).to(p.device))
        for m in (lifter, encoder):
            m.eval()                     # dropout off; autograd on
        head.train()                     # the training lattice / jitter / random background (drawn from the seeded generators)
        monkeypatch.setattr(bricks, 'FUSED_TRAINING', not route_b)
        monkeypatch.setattr(bricks, 'MERGED_OFF_LOGITS', not route_b)
        monkeypatch.setattr(bricks, 'VALUE_GRAD_SINK', not route_b)
        monkeypatch.setenv('SELFOCC_RB_SCATTER', 'atomic' if route_b else 'binned')
        head.model.field.fused_volume = not route_b
        if route_b:
            for m in encoder.modules():
                if isinstance(m, BEVCrossAttention):
                    m.camera_loop = False
        torch.manual_seed(11)
        np.random.seed(11)
        total, parts, out = hc.train_iteration(mods, cfg, hc.frame_inputs(cfg, name, D0, seed=5), global_iter=1)
        grads = {f'{tag}.{n}': p.grad.detach().clone() for tag, m in (('lifter', lifter), ('encoder', encoder), ('head', head))
                 for n, p in m.named_parameters() if p.grad is not None}
        res = (float(total), {k: float(v) for k, v in parts.items()}, grads)
        del mods, lifter, encoder, head, loss_fn, out, total
        torch.cuda.empty_cache()
        return res

    ta, pa, ga = run(False)
    tb, pb, gb = run(True)
    worst = dict(config=name, total=abs(ta - tb) / abs(tb), term=0.0, grad_l2=0.0, grad_name='')
    for k in pb:
        worst['term'] = max(worst['term'], abs(pa[k] - pb[k]) / max(abs(pb[k]), 1e-12))
    assert set(ga) == set(gb)
    for n in gb:
        e = float((ga[n].double() - gb[n].double()).norm() / gb[n].double().norm().clamp_min(1e-30))
        if e > worst['grad_l2']:
            worst.update(grad_l2=e, grad_name=n)
    try:
        import json
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "shipped_routes_parity.jsonl"), "a") as f:
            f.write(json.dumps(worst) + "\n")
    except OSError:
        pass
    # measured (profiles/r6_e_shipped_routes_parity.jsonl): every loss term <= 3.2e-7 relative; gradients 1.7e-5 .. 3.2e-3 rel-L2,
    # the worst always a `sampling_offsets` bias / the hw query plane of a 1-camera KITTI config: the two routes' projections
    # differ in the last bit, a sampling location within that of a pixel edge (a sample within it of a voxel face) lands on the
    # other side, and the bilinear / trilinear gradient is piece-wise constant there (the module-level test
    # tests/test_msda_gpu.py::test_merged_offset_logit_projection_module_equals_two_linears shows the same 1e-3 with random
    # weights and 1e-6 with dyadic ones).  A wrong route is O(1) off.
    assert worst['total'] <= 5e-6 and worst['term'] <= 5e-6, worst
    assert worst['grad_l2'] <= 2e-2, worst
