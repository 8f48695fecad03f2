"""CPU: host-side mirrors and the C oracle against golden vectors produced by the REAL
reference modules (tests/golden/make_golden.py; committed .npz fixtures)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from selfocc_amd.mapping import GridMeterMapping

G = os.path.join(os.path.dirname(__file__), "golden")
geo = np.load(os.path.join(G, "geometry.npz"))
los = np.load(os.path.join(G, "losses.npz"))

CFGS = {
    'occ': dict(nonlinear_mode='linear', h_size=[128, 0], h_range=[40.0, 0], h_half=False, w_size=[128, 0],
                w_range=[40.0, 0], w_half=False, d_size=[24, 0], d_range=[-1.0, 5.4, 5.4]),
    'kitti': dict(nonlinear_mode='linear', h_size=[256, 0], h_range=[51.2, 0], h_half=True, w_size=[128, 0],
                  w_range=[25.6, 0], w_half=False, d_size=[32, 0], d_range=[-2.0, 4.4, 4.4]),
    'twoseg': dict(nonlinear_mode='linear', h_size=[128, 32], h_range=[51.2, 28.8], h_half=False,
                   w_size=[128, 32], w_range=[51.2, 28.8], w_half=False, d_size=[20, 10], d_range=[-4.0, 4.0, 12.0]),
    'upscale': dict(nonlinear_mode='linear_upscale', h_size=[128, 32], h_range=[51.2, 28.8], h_half=False,
                    w_size=[128, 32], w_range=[51.2, 28.8], w_half=False, d_size=[20, 10], d_range=[-4.0, 4.0, 12.0]),
}


@pytest.mark.parametrize("name", list(CFGS))
def test_mapping_host_mirror_vs_reference(name):
    m = GridMeterMapping(**CFGS[name])
    assert [m.size_h, m.size_w, m.size_d] == geo[f'{name}.sizes'].tolist()
    xyz, grid = torch.tensor(geo[f'{name}.xyz']), torch.tensor(geo[f'{name}.grid'])
    assert torch.allclose(m.meter2grid(xyz), torch.tensor(geo[f'{name}.m2g']), rtol=1e-6, atol=1e-5)
    assert torch.allclose(m.meter2grid(xyz, True), torch.tensor(geo[f'{name}.m2g_norm']), rtol=1e-6, atol=1e-7)
    assert torch.allclose(m.grid2meter(grid), torch.tensor(geo[f'{name}.g2m']), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("name", ['occ', 'kitti', 'twoseg'])
def test_c_oracle_meter2grid_bit_exact_vs_reference(name):
    """the C oracle's canonical float32 order == the reference's torch evaluation, bit for bit"""
    m = GridMeterMapping(**CFGS[name])
    xyz = torch.tensor(geo[f'{name}.xyz'])
    assert torch.equal(oracle.meter2grid(m, xyz), torch.tensor(geo[f'{name}.m2g']))
    assert torch.equal(oracle.meter2grid(m, xyz, True), torch.tensor(geo[f'{name}.m2g_norm']))


def test_reference_kat_round_trip():
    """the reference's only in-tree known-answer test (mappings.py:312-318)"""
    m = GridMeterMapping(**CFGS['twoseg'])
    kat = torch.tensor(geo['kat.grid'])
    assert torch.allclose(m.grid2meter(kat), torch.tensor(geo['kat.meter']), rtol=1e-6, atol=1e-5)
    assert torch.allclose(m.meter2grid(m.grid2meter(kat)), kat, atol=1e-4)
    assert torch.allclose(torch.tensor(geo['kat.back']), kat, atol=1e-4)


def test_ray_sampler_vs_reference():
    from selfocc_amd.model.head import RaySampler
    assert torch.equal(RaySampler('fixed', [5, 8], [90, 160])(), torch.tensor(geo['rays.fixed']))
    np.random.seed(123)
    s = RaySampler('cellular', [6, 10], [96, 200], ray_upper_crop=8)
    got = np.stack([s().numpy() for _ in range(3)])
    assert np.allclose(got, geo['rays.cellular'], rtol=1e-6, atol=1e-4)


def test_point_sampling_and_ref_points_vs_reference():
    from selfocc_amd.model.encoder.utils import point_sampling, get_cross_view_ref_points
    assert torch.equal(get_cross_view_ref_points(5, 4, 3, [4, 4, 4]), torch.tensor(geo['cvref']))
    metas = [dict(lidar2img=geo['ps.lidar2img'], img_shape=(224, 400))]
    cam, mask = point_sampling(torch.tensor(geo['ps.ref3d']), metas)
    assert torch.equal(mask, torch.tensor(geo['ps.mask']))
    assert torch.allclose(cam, torch.tensor(geo['ps.cam']), rtol=1e-6, atol=1e-6)
    metas[0].update(focal_ratios_x=[1.0, 1.1, 0.9], focal_ratios_y=[1.0, 0.95, 1.05])
    cam2, _ = point_sampling(torch.tensor(geo['ps.ref3d']), metas)
    assert torch.allclose(cam2, torch.tensor(geo['ps.cam_focal']), rtol=1e-6, atol=1e-6)


def test_sh_degree0_colour_vs_reference():
    """SHRender(deg=0, relu) == relu(C0 * raw + 0.5): the formula the render kernel / oracle use"""
    from oracle import torch_port as tp
    assert torch.allclose(tp.sh0_color(torch.tensor(geo['sh.feat'])), torch.tensor(geo['sh.rgb']), rtol=1e-6, atol=1e-7)


def test_small_losses_vs_reference():
    from selfocc_amd.loss import RGBLossMS, SemCELossMS, SemLossMS, EdgeLoss3DMS
    R, S, Hi, Wi, rh, rw = los['dims'].tolist()
    rays, curr = torch.tensor(los['rays']), torch.tensor(los['curr'])
    colors = torch.tensor(los['colors'])
    v = RGBLossMS(1.0, [Hi, Wi], True, None)(dict(ms_colors=[colors], ms_rays=rays, gt_imgs=curr))
    assert torch.allclose(v, torch.tensor(los['rgb_l1.loss']), rtol=1e-6)
    v = RGBLossMS(1.0, [Hi, Wi], False, [rh, rw])(dict(ms_colors=[colors], ms_rays=rays, gt_imgs=curr))
    assert torch.allclose(v, torch.tensor(los['rgb_ssim.loss']), rtol=1e-5)
    sem, smeta = torch.tensor(los['sem']), [dict(sem=torch.tensor(los['semgt']))]
    v = SemCELossMS(1.0, [Hi, Wi], [rh, rw])(dict(sem=[sem], metas=smeta, ms_rays=rays))
    assert torch.allclose(v, torch.tensor(los['semce.loss']), rtol=1e-6)
    v = SemLossMS(1.0, [Hi, Wi], [rh, rw])(dict(sem=[sem], metas=smeta, ms_rays=rays))
    assert torch.allclose(v, torch.tensor(los['sembce.loss']), rtol=1e-6)
    v = EdgeLoss3DMS(1.0, None, img_size=[Hi, Wi], ray_resize=[rh, rw])(
        dict(curr_imgs=curr, ms_depths=[torch.tensor(los['depth'])], ms_rays=rays))
    assert torch.allclose(v, torch.tensor(los['edge.loss']), rtol=1e-5)


def test_reproj_port_vs_reference_loss_no_ssim():
    """The torch port used as the kernel's oracle, assembled into the full no-SSIM loss, equals
    the real reference class (value and gradient): pins oracle/torch_port.reproj_sample_port."""
    from oracle import torch_port as tp
    R, S, Hi, Wi, rh, rw = los['dims'].tolist()
    rays = torch.tensor(los['rays'])
    tot, grads = 0., []
    for cam in range(2):
        w = torch.tensor(los['weights'][cam]).reshape(R, S).requires_grad_(True)
        curr_rgb = torch.nn.functional.grid_sample(
            torch.tensor(los['curr'][0, cam])[None], (rays / torch.tensor([Wi, Hi]) * 2 - 1).reshape(1, 1, -1, 2),
            mode='bilinear', padding_mode='border', align_corners=True).reshape(3, -1).T
        l1, comb, anyv = tp.reproj_sample_port(w, torch.tensor(los['ts'][cam]).reshape(R, S),
                                               torch.tensor(los['deltas'][cam]).reshape(R, S), rays, curr_rgb,
                                               torch.tensor(los['img2prevImg'][cam], dtype=torch.float32),
                                               torch.tensor(los['img2nextImg'][cam], dtype=torch.float32),
                                               torch.tensor(los['prev'][0, cam]), torch.tensor(los['next'][0, cam]),
                                               float(Hi), float(Wi))
        def samp(img):
            return torch.nn.functional.grid_sample(img[None], (rays / torch.tensor([Wi, Hi]) * 2 - 1).reshape(1, 1, -1, 2),
                                                   mode='bilinear', padding_mode='border', align_corners=True).reshape(3, -1).T
        pn = torch.where(anyv > 0, l1, torch.full_like(l1, 1e3))
        mp_ = (samp(torch.tensor(los['prev'][0, cam])) - curr_rgb).abs().mean(-1)
        mn_ = (samp(torch.tensor(los['next'][0, cam])) - curr_rgb).abs().mean(-1)
        loss = torch.stack([pn, mp_, mn_], -1).min(-1)[0].mean()
        loss.backward()
        tot = tot + loss.detach()
        grads.append(w.grad.flatten() / 2)
    assert torch.allclose(tot / 2, torch.tensor(los['combine_nossim_deltas.loss']), rtol=1e-5)
    assert torch.allclose(torch.stack(grads), torch.tensor(los['combine_nossim_deltas.gw']), rtol=1e-3, atol=1e-7)


@pytest.mark.skipif(not os.path.isdir("/root/reference/config"), reason="reference configs only exist in the build container")
def test_shipped_configs_parse_and_resolve():
    """every shipped experiment config parses unchanged and every hot-path ``type`` resolves"""
    import glob
    from selfocc_amd.config import Config
    from selfocc_amd.registry import MODELS, OPENOCC_LOSS
    import selfocc_amd.model, selfocc_amd.loss  # noqa: F401
    files = sorted(glob.glob("/root/reference/config/*/*.py"))
    assert len(files) >= 7
    for f in files:
        if '_base_' in f:
            continue
        cfg = Config.fromfile(f)
        for part in ('lifter', 'encoder', 'head'):
            assert cfg.model[part]['type'] in MODELS, (f, part)
        for lc in cfg.loss['loss_cfgs']:
            assert lc['type'] in OPENOCC_LOSS, (f, lc['type'])
        assert cfg.loss['type'] in OPENOCC_LOSS

        def walk(d):
            if isinstance(d, dict):
                if 'type' in d and d['type'] in ('TPVFormerLayer', 'BEVFormerLayer', 'TPVCrossAttention',
                                                 'BEVCrossAttention', 'BEVDeformableAttention',
                                                 'CrossViewHybridAttention', 'MultiScaleDeformableAttention',
                                                 'TPVPositionalEncoding', 'BEVPositionalEncoding'):
                    assert d['type'] in MODELS
                for v in d.values():
                    walk(v)
            elif isinstance(d, (list, tuple)):
                for v in d:
                    walk(v)
        walk(cfg.model['encoder'])
