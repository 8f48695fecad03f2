"""Shared by scripts/bench_hotpath_*.py: the hot path BUILT FROM THE SHIPPED CONFIGS (scripts/shipped_cfg/*.json, dumped from
config/**/*.py by scripts/dump_shipped_configs.py through selfocc_amd.config) via the registries — the same code path
INTEGRATION.md's drop-in uses — plus the synthetic stand-ins for what is out of scope (camera rig, FPN feature maps, images).
Eval frames apply the reference's own eval-time overrides (utils/config_tools.py:10-14, 63-67, 90-92: NUM_RAYS, fixed lattice,
trans_kw)."""
import copy
import json
import math
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NUM_RAYS = {'nuscenes': [450, 800], 'kitti': [176, 608]}          # utils/config_tools.py:1-8


def shipped(name):
    return json.load(open(os.path.join(ROOT, "scripts", "shipped_cfg", name + ".json")))


def modify_for_eval(cfg, dataset, novel_depth=False):
    """utils/config_tools.py:10-14, 56-67, 88-92, the keys the hot path reads"""
    cfg = copy.deepcopy(cfg)
    nr = NUM_RAYS[dataset]
    cfg['num_rays'] = nr
    cfg['loss']['loss_cfgs'][0]['ray_resize'] = nr
    cfg['loss']['loss_cfgs'][1]['ray_resize'] = nr
    cfg['model']['head'].update(ray_sample_mode='fixed', ray_number=nr, trans_kw='img2lidar')
    if novel_depth:
        cfg['model']['head'].update(trans_kw='render_img2lidar')
    return cfg


def build(cfg, device, want_loss=False):
    from selfocc_amd.registry import MODELS, OPENOCC_LOSS
    import selfocc_amd.model, selfocc_amd.loss  # noqa: F401
    m = cfg['model']
    lifter = MODELS.build(copy.deepcopy(m['lifter'])).to(device)
    encoder = MODELS.build(copy.deepcopy(m['encoder'])).to(device)
    encoder.init_weights()
    head = MODELS.build(copy.deepcopy(m['head'])).to(device)
    loss = OPENOCC_LOSS.build(copy.deepcopy(cfg['loss'])) if want_loss else None
    return lifter, encoder, head, loss


def ring_cameras(n, img_hw, focal, z=1.5):
    """n pinholes at yaw steps of 360 / n around the ego origin: (img2lidar (n, 4, 4), lidar2img (n, 4, 4), K)"""
    Hh, Ww = img_hw
    K = np.array([[focal, 0, Ww / 2.0, 0], [0, focal, Hh / 2.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    c2w, l2i = [], []
    for i in range(n):
        yaw = math.radians(360.0 / n * i + (90.0 if n == 1 else 0.0))      # the KITTI camera looks along +y (the box is in front)
        fwd = np.array([math.cos(yaw), math.sin(yaw), 0.0]); right = np.array([math.sin(yaw), -math.cos(yaw), 0.0])
        down = np.array([0, 0, -1.0])
        m = np.eye(4); m[:3, :3] = np.stack([right, down, fwd], 1); m[:3, 3] = [0.2 * i, 0.1 + (0.5 if n == 1 else 0.0), z]
        c2w.append(m @ np.linalg.inv(K)); l2i.append(K @ np.linalg.inv(m))
    return np.stack(c2w), np.stack(l2i), K


def motion(K, yaw, tx, tz):
    y = np.deg2rad(yaw)
    Rm = np.array([[np.cos(y), 0, np.sin(y), tx], [0, 1, 0, 0], [-np.sin(y), 0, np.cos(y), tz], [0, 0, 0, 1]])
    return K @ Rm @ np.linalg.inv(K)


def fpn_feats(n_cams, dim, img_hw, device, strides=(8, 16, 32, 64)):
    """random stand-ins for the FPN outputs: ceil(img / stride) per level"""
    return [torch.randn(1, n_cams, dim, -(-img_hw[0] // s), -(-img_hw[1] // s), device=device) for s in strides]


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


# ---- every shipped experiment config (config/{nuscenes,kitti,kitti_raw}/*.py): one training iteration + its evaluation entry ----
# dataset = the `dataset` argument the reference's entry scripts hand to utils/config_tools.py:modify_for_eval; eval = which entry
# script the docs pair with the config (docs/get_started.md:17-107):
#   render        eval_depth.py:150-227           head.prepare + head.render (return_max_depth=True, eval_depth.py:70)
#   render_novel  eval_novel_depth[_kitti].py     the same with trans_kw = render_img2lidar (return_max_depth=True, :70)
#   occ3d         eval_iou.py:166-294             head.forward_occ + Occ3D resample / LUT / MeanIoU counts
#   occ_kitti     eval_iou_kitti.py:153-200       head.forward_occ(aabb, resolution 0.2) + threshold / crops / IoU counts
SHIPPED = {
    'nuscenes_occ': dict(dataset='nuscenes', eval='occ3d', focal=1266.0, cam_z=1.5),
    'nuscenes_occ_bev': dict(dataset='nuscenes', eval='occ3d', focal=1266.0, cam_z=1.5),
    'nuscenes_depth': dict(dataset='nuscenes', eval='render', focal=1266.0, cam_z=1.5),
    'nuscenes_novel_depth': dict(dataset='nuscenes', eval='render_novel', focal=1266.0, cam_z=1.5),
    'kitti_occ': dict(dataset='kitti', eval='occ_kitti', focal=707.0, cam_z=1.7),
    'kitti_novel_depth': dict(dataset='kitti', eval='render_novel', focal=707.0, cam_z=1.7),
    'kitti_raw_depth': dict(dataset='kitti_raw', eval='render', focal=707.0, cam_z=1.7),
}
NUM_RAYS['kitti_raw'] = [176, 608]
KITTI_OCC_AABB = [-25.6, 0, -2.0, 25.6, 51.2, 4.4]            # eval_iou_kitti.py:161


def shipped_for_eval(name):
    info = SHIPPED[name]
    cfg = modify_for_eval(shipped(name), info['dataset'], novel_depth=info['eval'] == 'render_novel')
    if info['eval'] in ('render', 'render_novel'):
        cfg['model']['head']['return_max_depth'] = True          # eval_depth.py:70, eval_novel_depth.py:70, eval_novel_depth_kitti.py:70
    return cfg


def frame_inputs(cfg, name, device, seed=0, want_images=True):
    """One synthetic frame for a shipped config: the metas the hot path reads as host numpy, FPN maps,
    and — for training — the four image stacks train.py:205-208 moves to the device (+ the OpenSeeD label map when a
    semantic loss is configured, train.py:214-215).  ``seed`` moves the rig so that no frame repeats another's matrices."""
    info, m = SHIPPED[name], cfg['model']
    n_cams = m['encoder']['num_cams']
    img = tuple(cfg['img_size'])
    ray_img = tuple(m['head']['ray_img_size'])
    c2w, l2i, K = ring_cameras(n_cams, ray_img, info['focal'], z=info['cam_z'] + 0.01 * seed)
    novel = c2w.copy()
    novel[:, 1, 3] += 1.0
    metas = [dict(lidar2img=l2i, img2lidar=c2w, temImg2lidar=c2w, render_img2lidar=novel, img_shape=img,
                  img2prevImg=np.stack([motion(K, 2, 0.3, -0.8)] * n_cams),
                  img2nextImg=np.stack([motion(K, -2, -0.3, 0.8)] * n_cams))]
    g = torch.Generator(device='cpu').manual_seed(1000 + seed)
    feats = [torch.randn(1, n_cams, m['encoder']['embed_dims'], -(-img[0] // s), -(-img[1] // s), generator=g).to(device)
             for s in (8, 16, 32, 64)]
    imgs = None
    if want_images:
        low = {k: torch.rand(n_cams, 3, 48, 100, generator=g) for k in ('curr_imgs', 'prev_imgs', 'next_imgs', 'color_imgs')}
        imgs = {k: torch.nn.functional.interpolate(v.to(device), size=ray_img, mode='bilinear', align_corners=True)[None].contiguous()
                for k, v in low.items()}
        if any(c['type'] in ('SemLossMS', 'SemCELossMS') for c in cfg['loss']['loss_cfgs']):
            n_sem = m['head']['color_dims'] - 3
            metas[0]['sem'] = torch.randint(0, n_sem, (n_cams, *ray_img), generator=g).to(device)
    return metas, feats, imgs


def train_iteration(mods, cfg, fr, global_iter=0, events=None):
    """train.py:219-239 on the hot path: lifter -> encoder -> head.forward -> loss_input_convertion -> MultiLoss -> backward"""
    lifter, encoder, head, loss_fn = mods
    metas, feats, imgs = fr
    mark = (lambda k: events.__setitem__(k, ev())) if events is not None else (lambda k: None)
    mark('t0')
    rep = encoder(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
    mark('t1')
    out = head(rep, metas, global_iter=global_iter)
    mark('t2')
    loss_input = dict(metas=metas, curr_feats=imgs['curr_imgs'], prev_feats=imgs['prev_imgs'], next_feats=imgs['next_imgs'], **imgs)
    for k, v in cfg['loss_input_convertion'].items():
        loss_input[k] = out[v]
    total, parts = loss_fn(loss_input)
    mark('t3')
    total.backward()
    mark('t4')
    return total, parts, out


_OCC3D_GRID = {}


def eval_entry(mods, cfg, name, fr, state=None, events=None):
    """the evaluation entry the docs pair with the config (see SHIPPED); returns a dict of its results"""
    from selfocc_amd.occ import occ_resample, MeanIoU, OPENSEED2NUSCENES
    lifter, encoder, head = mods[:3]
    metas, feats, _ = fr
    kind = SHIPPED[name]['eval']
    d = feats[0].device
    mark = (lambda k: events.__setitem__(k, ev())) if events is not None else (lambda k: None)
    mark('t0')
    rep = encoder(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
    mark('t1')
    if kind in ('render', 'render_novel'):
        head.prepare(rep, metas)
        mark('t2')
        out = head.render(metas, batch=90000)                       # README.md:92 `--batch 90000`
        mark('t3')
        return out
    if kind == 'occ3d':
        res = head.forward_occ(rep, metas, aabb=cfg['model']['head']['roi_aabb'], resolution=0.4)      # eval_iou.py: --resolution 0.4
        mark('t2')
        key = str(d)
        if key not in _OCC3D_GRID:      # ego -> lidar resampling coordinates of the Occ3D grid (eval_iou.py:211-232), a constant here
            _OCC3D_GRID[key] = torch.stack(torch.meshgrid(torch.linspace(0.015, 0.985, 200), torch.linspace(0.02, 0.99, 200),
                                                          torch.linspace(0.05, 0.95, 16), indexing='ij'), -1).to(d).contiguous()
        got = occ_resample(res['sdf'], _OCC3D_GRID[key], 0.0, logits=res['logits'], lut=OPENSEED2NUSCENES, crop=(6, 6, 6, 6, 0, 4))
        if state is not None:
            if 'miou' not in state:
                cls = list(range(1, 17))
                state['miou'] = MeanIoU(cls, 0, [str(c) for c in cls], True, 0)
                state['miou'].reset()
                gg = torch.Generator(device='cpu').manual_seed(5)
                state['gt'] = torch.randint(0, 18, tuple(got['sem'].shape), generator=gg).to(d).int()
                state['mask'] = (torch.rand(tuple(got['sem'].shape), generator=gg) > 0.3).to(d)
            state['miou']._after_step(got['sem'], state['gt'], state['mask'])
        mark('t3')
        return dict(res, occ=got['occ'], sem_nus=got['sem'])
    # occ_kitti: eval_iou_kitti.py:153-185
    res = head.forward_occ(rep, metas, aabb=KITTI_OCC_AABB, resolution=0.2)
    mark('t2')
    pred = (res['sdf'] <= 0).to(torch.int)
    pred[..., 28:] = 0
    pred[-6:, ...] = 0
    pred[:, :6, :] = 0
    pred[:, -6:, :] = 0
    if state is not None:
        if 'miou' not in state:
            state['miou'] = MeanIoU([1], 0, ['occupied'], False, 0)
            state['miou'].reset()
            gg = torch.Generator(device='cpu').manual_seed(5)
            state['gt'] = (torch.rand(tuple(pred.shape), generator=gg) > 0.8).to(d).int()
        state['miou']._after_step(pred, state['gt'])
    mark('t3')
    return dict(res, occ=pred)
