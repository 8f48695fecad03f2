"""GPU: SURVEY section 8 row f-4 — the hot path issues no host <-> device synchronisation.

After one warm-up pass (lazy workspaces, constant shape tensors, hipBLASLt handles) a NEW frame — new camera
matrices, new features — runs under ``torch.cuda.set_sync_debug_mode("error")``, which raises on every synchronising
call torch knows about (``.item()``, ``nonzero()``, blocking host copies ...):
  * a depth-evaluation frame: lifter -> encoder -> head.prepare -> head.render      (eval_depth.py:150-227)
  * an occupancy-evaluation frame: ... -> head.forward_occ                           (eval_iou.py:166-294)
  * a training step: ... -> head.forward -> MultiLoss -> backward                    (train.py:198-254)
The reference syncs ~20 x per iteration on this path (6 ``nonzero()`` per plane per layer in BEVCrossAttention
:91-94, one ``.item()`` per loss in MultiLoss :33-41, ``.cpu()`` in the max-depth :430-438, per-call matrix uploads).
"""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D0 = torch.device("cuda:0")


def _stages(train):
    import test_head_gpu as th
    from selfocc_amd.registry import MODELS, OPENOCC_LOSS
    import selfocc_amd.model, selfocc_amd.loss  # noqa: F401
    dim, H, W, Z = 32, 32, 32, 4
    layer = dict(type='TPVFormerLayer',
                 attn_cfgs=[dict(type='CrossViewHybridAttention', embed_dims=dim, num_heads=2, num_levels=3,
                                 num_points=4, dropout=0.1, batch_first=True),
                            dict(type='TPVCrossAttention', embed_dims=dim, num_cams=2, dropout=0.1, batch_first=True,
                                 num_heads=2, num_levels=2, num_points=[3, 3, 2])],
                 feedforward_channels=2 * dim, ffn_dropout=0.1,
                 operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'))
    enc_cfg = dict(type='TPVFormerEncoder', mapping_args=th.MAP, embed_dims=dim, num_cams=2, num_feature_levels=2,
                   positional_encoding=dict(type='TPVPositionalEncoding', num_freqs=[3] * 3, embed_dims=dim,
                                            tot_range=[0.0, 0.0, -1.0, 12.8, 12.8, 2.0]),
                   num_points_cross=[3, 3, 2], num_points_self=[4] * 3, transformerlayers=[layer, layer], num_layers=2)
    lifter = MODELS.build(dict(type='TPVQueryLifter', tpv_h=H, tpv_w=W, tpv_z=Z, dim=dim)).to(D0)
    enc = MODELS.build(copy.deepcopy(enc_cfg)).to(D0)
    head = th.make_head(color_dims=8, return_sem=True, ray_sample_mode='cellular' if train else 'fixed',
                        render_bkgd='random' if train else 'white', return_uniform_sdf=train)
    loss_fn = OPENOCC_LOSS.build(dict(type='MultiLoss', loss_cfgs=[
        dict(type='ReprojLossMonoMultiNewCombine', weight=1.0, no_ssim=False, img_size=[64, 64], ray_resize=[6, 10],
             input_dict={'curr_imgs': 'curr_imgs', 'prev_imgs': 'prev_imgs', 'next_imgs': 'next_imgs',
                         'ray_indices': 'ray_indices', 'weights': 'weights', 'ts': 'ts', 'metas': 'metas', 'ms_rays': 'ms_rays'}),
        dict(type='RGBLossMS', weight=0.1, img_size=[64, 64], no_ssim=False, ray_resize=[6, 10],
             input_dict={'ms_colors': 'ms_colors', 'ms_rays': 'ms_rays', 'gt_imgs': 'curr_imgs'}),
        dict(type='EikonalLoss', weight=0.1), dict(type='SecondGradLoss', weight=0.01),
        dict(type='EdgeLoss3DMS', weight=0.01, img_size=[64, 64], ray_resize=[6, 10]),
        dict(type='SoftSparsityLoss', weight=0.005, input_dict={'density': 'uniform_sdf'})]))
    for m in (lifter, enc, head):
        m.train(train)
    return th, lifter, enc, head, loss_fn


def _frame(th, seed):
    """a frame = camera matrices (host numpy, as the dataset hands them over) + image features + images"""
    _, metas, imgs = th.make_inputs(seed=seed)
    for k in ('img2lidar', 'temImg2lidar'):
        m = np.array(metas[0][k], dtype=np.float64)
        m[:, :3, 3] += 0.01 * (seed + 1)                 # every frame has its own matrices: no cache hit on contents
        metas[0][k] = m
    metas[0]['lidar2img'] = np.stack([np.linalg.inv(m) for m in metas[0]['img2lidar']])
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(1, 2, 32, 8, 8, generator=g).to(D0), torch.randn(1, 2, 32, 4, 4, generator=g).to(D0)]
    return metas, feats, imgs


class no_sync:
    def __enter__(self):
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")

    def __exit__(self, *a):
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()


def test_depth_and_occupancy_eval_frames_are_sync_free(hip):
    th, lifter, enc, head, _ = _stages(train=False)
    os.environ['eval'] = 'true'
    try:
        def frame(metas, feats, _imgs):
            rep = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
            head.prepare(rep, metas)
            out = head.render(metas, batch=90000)
            occ = head.forward_occ(rep, metas, aabb=th.AABB, resolution=0.4)
            return out, occ
        with torch.no_grad():
            frame(*_frame(th, 0))                      # warm-up: workspaces, constant tensors
            new = _frame(th, 1)                        # the data loader's side (images / features .cuda(), train.py:204-208)
            with no_sync():
                out, occ = frame(*new)                 # a new frame: nothing on the host waits for the device
        assert torch.isfinite(out['ms_depths'][0]).all() and out['ms_max_depths'][0].shape == (1, 2, 60)
        assert occ['sdf'].shape == (32, 32, 7) and occ['sem'].dtype == torch.int64
    finally:
        os.environ['eval'] = 'false'


def test_training_step_is_sync_free(hip):
    th, lifter, enc, head, loss_fn = _stages(train=True)
    os.environ['eval'] = 'false'
    params = [p for m in (lifter, enc, head) for p in m.parameters()]

    def step(metas, feats, imgs):
        rep = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        out = head(rep, metas, global_iter=7)
        total, parts = loss_fn(dict(out, metas=metas, **imgs))
        total.backward()
        return total, parts
    np.random.seed(0)
    step(*_frame(th, 0))
    for p in params:
        p.grad = None
    new = _frame(th, 1)
    with no_sync():
        total, parts = step(*new)
    assert torch.isfinite(total).all() and len(parts) == 6
    # the per-loss values stayed on the device; they still print like the reference's floats (train.py:260-263)
    assert all(isinstance(v, torch.Tensor) and v.is_cuda for v in parts.values())
    line = ', '.join(f'{k}: {v:.5f}' for k, v in parts.items())
    assert 'EikonalLoss: ' in line
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in params)
