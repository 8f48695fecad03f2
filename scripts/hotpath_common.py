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
