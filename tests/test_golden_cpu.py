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
    # (the SSIM variant runs the HIP kernel: tests/test_golden_gpu.py::test_rgb_ssim_loss_vs_reference_class)
    with pytest.raises(RuntimeError):
        RGBLossMS(1.0, [Hi, Wi], False, [rh, rw])(dict(ms_colors=[colors], ms_rays=rays, gt_imgs=curr))
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


@pytest.mark.skipif(not os.path.isdir("/root/reference/config"), reason="reference configs only exist in the build container")
def test_dumped_shipped_configs_are_what_the_reference_ships():
    """scripts/shipped_cfg/*.json (what the GPU box builds the seven configs from, tests/test_shipped_configs_gpu.py and bench.py's
    hot_path) == the hot-path part of config/{nuscenes,kitti,kitti_raw}/*.py as selfocc_amd.config reads them today: all seven, no
    hand edits (scripts/dump_shipped_configs.py regenerates them byte for byte)."""
    import glob
    import json
    from selfocc_amd.config import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(f for d in ("nuscenes", "kitti", "kitti_raw") for f in glob.glob(f"/root/reference/config/{d}/*.py"))
    dumps = sorted(glob.glob(os.path.join(root, "scripts", "shipped_cfg", "*.json")))
    assert len(files) == len(dumps) == 7
    for f in files:
        name = os.path.basename(f)[:-3]
        got = json.load(open(os.path.join(root, "scripts", "shipped_cfg", name + ".json")))
        c = Config.fromfile(f).to_dict()
        assert got["source"] == os.path.relpath(f, "/root/reference")
        want = {"model": {k: c["model"][k] for k in ("type", "lifter", "encoder", "head")}, "loss": c["loss"],
                "loss_input_convertion": c["loss_input_convertion"], "img_size": c["img_size"], "num_rays": c["num_rays"],
                "optimizer": c["optimizer"], "grad_max_norm": c["grad_max_norm"], "amp": c.get("amp", False)}
        if "crop_size" in c:
            want["crop_size"] = c["crop_size"]
        # through JSON: tuples become lists on both sides
        assert json.loads(json.dumps(want, sort_keys=True)) == {k: got[k] for k in want}, name


# ---- fixtures of tests/golden/more.npz: Img2LiDAR, BEVNeRF, the small volume losses, LUT ----------------------
mor = np.load(os.path.join(G, "more.npz"))


def test_img2lidar_vs_reference():
    """Img2LiDAR (model/head/nerfacc_head/img2lidar.py:25-70): single key, two-split, novel view, eval key."""
    from selfocc_amd.model.head.neus_head import Img2LiDAR
    metas = [dict(img2lidar=list(mor['i2l.img2lidar']), temImg2lidar=list(mor['i2l.temImg2lidar']))]
    rays = torch.tensor(mor['i2l.rays'])
    os.environ['eval'] = 'false'
    cases = {'single': Img2LiDAR('img2lidar'), 'split': Img2LiDAR(['img2lidar', 'temImg2lidar']),
             'novel': Img2LiDAR('temImg2lidar', novel_view=[0.5, -1.0, 0.25, 12.0])}
    for name, mod in cases.items():
        o, d = mod(metas, rays)
        assert torch.allclose(o, torch.tensor(mor[f'i2l.{name}.origin']), rtol=1e-6, atol=1e-6), name
        assert torch.allclose(d, torch.tensor(mor[f'i2l.{name}.dir']), rtol=1e-5, atol=1e-6), name
    os.environ['eval'] = 'true'
    try:
        o, d = Img2LiDAR('temImg2lidar', trans_kw_eval=['img2lidar'])(metas, rays)
    finally:
        os.environ['eval'] = 'false'
    assert torch.allclose(o, torch.tensor(mor['i2l.eval.origin']), rtol=1e-6, atol=1e-6)
    assert torch.allclose(d, torch.tensor(mor['i2l.eval.dir']), rtol=1e-5, atol=1e-6)


def test_in_kernel_pixel_ray_vs_reference_img2lidar():
    """The ray the render kernel generates from a lattice (so_pixel_ray; restated by the C oracle) == the
    reference's Img2LiDAR direction, normalised as neus_head.py:321-327 does."""
    from selfocc_amd import synthetic as sy
    from selfocc_amd.render import RaySet
    M = torch.tensor(mor['i2l.img2lidar'], dtype=torch.float32)
    rs = RaySet(img2lidar=M, nx=8, ny=5, sx=400 / 8, sy=224 / 5, ox=1.5, oy=0.75)
    ex = sy.explicit_rays(rs)                       # the torch restatement used by the tests
    from selfocc_amd.model.head.neus_head import Img2LiDAR, RaySampler
    pix = RaySampler.pixels(5, 8, 400 / 8, 224 / 5, 1.5, 0.75, torch.device('cpu'))
    o, d = Img2LiDAR('img2lidar')([dict(img2lidar=list(mor['i2l.img2lidar']))], pix)
    d = d.flatten(0, 2)
    dn = d.norm(dim=-1, keepdim=True)
    assert torch.allclose(ex.dirs, d / dn, rtol=1e-6, atol=1e-7) and torch.allclose(ex.dir_norm, dn[:, 0], rtol=1e-6)
    assert torch.allclose(ex.origins, o[0].unsqueeze(1).expand(-1, 40, -1).flatten(0, 1), atol=1e-7)


def test_volume_losses_vs_reference():
    """EikonalLoss / SecondGradLoss / the sparsity family (loss/eikonal_loss.py, second_grad_loss.py,
    sparsity_loss.py:7-113) on the reference's own outputs."""
    from selfocc_amd.registry import OPENOCC_LOSS
    import selfocc_amd.loss  # noqa: F401
    t = lambda k: torch.tensor(mor[k])
    B = OPENOCC_LOSS.build
    close = lambda a, k: torch.allclose(a, t(k), rtol=1e-5, atol=1e-8)
    assert close(B(dict(type='EikonalLoss', weight=0.1))(dict(eik_grad=t('vl.eik_grad'))), 'vl.eikonal')
    assert close(B(dict(type='SecondGradLoss', weight=0.01))(dict(second_grad=t('vl.second_grad'))), 'vl.second')
    assert close(B(dict(type='SparsityLoss', weight=0.5, scale=0.7))(dict(density=t('vl.density'))), 'vl.sparsity')
    assert close(B(dict(type='SoftSparsityLoss', weight=0.005, input_dict={'density': 'uniform_sdf'}))(
        dict(uniform_sdf=t('vl.density'))), 'vl.soft_sparsity')
    assert close(B(dict(type='HardSparsityLoss', weight=1.0, scale=2.0, thresh=0.3, crop=[[1, 2], [0, 1], [0, 0]]))(
        dict(density=t('vl.density').clone())), 'vl.hard_sparsity')
    ts, sdfs = list(t('vl.ts')), list(t('vl.sdfs'))
    assert close(B(dict(type='AdaptiveSparsityLoss', weight=1.0, slack=4.0))(
        dict(sdfs=sdfs, ts=ts, ms_depths=[t('vl.depths')])), 'vl.adaptive_sparsity')


@pytest.mark.parametrize("tag", ['tpv', 'bev'])
def test_sdf_field_volume_vs_reference_bevnerf(tag):
    """SDFField.pre_compute_density_color == the authors' in-repo field BEVNeRF (bev_nerf.py:62-95) with ITS
    state dict loaded by name; then the C oracle's trilinear lookup == BEVNeRF's grid_sample lookup, bit for bit."""
    from selfocc_amd.model.head.neus_head import SDFField
    mapping_args = dict(nonlinear_mode='linear', h_size=[4, 0], h_range=[8.0, 0], h_half=False, w_size=[3, 0],
                        w_range=[6.0, 0], w_half=False, d_size=[2, 0], d_range=[-1.0, 3.0, 3.0])
    f = SDFField(mapping_args, embed_dims=16, color_dims=7, density_layers=2, sh_deg=0, tpv=(tag == 'tpv'))
    sd = {k[len(f'bevnerf.{tag}.sd.'):]: torch.tensor(mor[k]) for k in mor.files if k.startswith(f'bevnerf.{tag}.sd.')}
    missing, unexpected = f.load_state_dict(sd, strict=False)
    assert unexpected == [] and missing == ['variance']          # BEVNeRF has no NeuS variance parameter
    rep = [torch.tensor(mor[f'bevnerf.{tag}.rep{i}']) for i in range(3)]
    with torch.no_grad():
        vol = f.pre_compute_density_color(rep if tag == 'tpv' else rep[0])
    ref_vol = torch.tensor(mor[f'bevnerf.{tag}.volume'])          # (1, 8, H, W, D)
    assert torch.allclose(vol.to_reference_layout(), ref_vol, rtol=1e-5, atol=1e-6)
    xyz = torch.tensor(mor[f'bevnerf.{tag}.xyz'])
    sdf, _ = oracle.field_sdf(vol.mapping, ref_vol[0, 0].contiguous(), xyz)
    assert torch.equal(sdf, torch.tensor(mor[f'bevnerf.{tag}.lookup'])[:, 0])


def test_openseed2nuscenes_lut_vs_reference():
    from selfocc_amd.occ import OPENSEED2NUSCENES
    assert OPENSEED2NUSCENES == mor['iou.openseed2nuscenes'].tolist()


def test_our_stages_bind_the_reference_segmentor_keywords():
    """Every keyword set the reference's TPVSegmentor passes to lifter / encoder / head (recorded from the real class
    into segmentor_protocol.json) binds to the corresponding method signature of OUR classes — the CPU half of the
    drop-in check (the GPU half replays the calls with tensors)."""
    import inspect
    import selfocc_amd.model  # noqa: F401
    from selfocc_amd.registry import MODELS, HEADS
    proto = json.load(open(os.path.join(G, "segmentor_protocol.json")))
    cls = dict(lifter=[MODELS.get('TPVQueryLifter'), MODELS.get('BEVQueryLifter')],
               encoder=[MODELS.get('TPVFormerEncoder'), MODELS.get('BEVFormerEncoder')], head=[HEADS.get('NeuSHead')])
    assert set(proto) == {'train', 'prepare', 'occ_only'}
    for mode, rec in proto.items():
        assert [c['stage'] for c in rec['calls']] == ['lifter', 'encoder', 'head']
        for call in rec['calls']:
            for c in cls[call['stage']]:
                assert c is not None
                sig = inspect.signature(getattr(c, call['method']))
                sig.bind(None, **{k: None for k in call['kwargs']})     # raises TypeError if a keyword does not bind


@pytest.mark.parametrize("tag,mode", [('tpv', 'train'), ('bev', 'train'), ('tpv', 'evalfwd'), ('bev', 'evalfwd')])
def test_head_ray_construction_vs_reference_head(tag, mode):
    """Host half of NeuSHead (no kernel involved): pixel lattice, origin / direction / direction_norm and the
    get_uniform_sdf lattice exactly as the REAL neus_head.py built them (tests/golden/head.npz; :265-281, :508-527)."""
    import copy
    import json
    from selfocc_amd.registry import MODELS
    from selfocc_amd.occ import uniform_lattice
    import selfocc_amd.model  # noqa: F401
    z = np.load(os.path.join(G, "head.npz"))
    cfg = json.load(open(os.path.join(G, "head_cfg.json")))[tag]
    head = MODELS.build(dict(type='NeuSHead', **copy.deepcopy(cfg)))
    metas = [dict(img2lidar=list(z[f'{tag}.img2lidar']), temImg2lidar=list(z[f'{tag}.temImg2lidar']))]
    os.environ['eval'] = 'true' if mode == 'evalfwd' else 'false'
    try:
        np.random.seed(77)
        rays, pix, num_cams, num_rays = head._rays(metas, torch.device('cpu'))
        origin, direction = head.img2lidar(metas, pix)
    finally:
        os.environ['eval'] = 'false'
    direction = direction.flatten(0, 2)
    dn = torch.norm(direction, dim=-1, keepdim=True)
    t = lambda k: torch.tensor(z[f'{tag}.{mode}.{k}'])
    assert torch.allclose(pix, t('ms_rays'), rtol=1e-6, atol=1e-5)
    assert torch.allclose(origin.unsqueeze(2).repeat(1, 1, num_rays, 1).flatten(0, 2), t('origin'), rtol=1e-6, atol=1e-6)
    assert torch.allclose(direction / dn, t('direction'), rtol=1e-5, atol=1e-6)
    assert torch.allclose(dn, t('direction_norm'), rtol=1e-5, atol=1e-6)
    assert num_cams * num_rays == t('origin').shape[0]
    # the in-kernel lattice descriptor generates the same pixels
    if rays.pixel_grid:
        xs = torch.arange(rays.nx, dtype=torch.float32) * np.float32(rays.sx) + np.float32(rays.ox)
        ys = torch.arange(rays.ny, dtype=torch.float32) * np.float32(rays.sy) + np.float32(rays.oy)
        lat = torch.stack([xs[None].expand(rays.ny, -1), ys[:, None].expand(-1, rays.nx)], -1).flatten(0, 1)
        assert torch.allclose(lat, t('ms_rays'), rtol=1e-6, atol=1e-4)
    xyz = uniform_lattice([-6.0, -5.0, -0.5, 6.0, 7.0, 2.5], 0.5, 'cpu')
    assert torch.equal(xyz, torch.tensor(z[f'{tag}.occ.xyz']))
    assert torch.equal(uniform_lattice(cfg['roi_aabb'], cfg['resolution'], 'cpu'), torch.tensor(z[f'{tag}.occdef.xyz']))


def test_encoder_full_fixture_regenerates_from_seeds():
    """tests/golden/encoder_full.npz stores gradients but not the 1.8 M parameters / inputs they belong to: both sides
    regenerate those from seeds.  Here (no GPU): our modules build at the shipped structure, take the reference's
    parameter names in the same sorted order, and the regenerated tensors match the generator's checksums."""
    from test_golden_encoder_full_gpu import _setup_cpu
    z, spec, enc, lifter, feats, l2i, loss_dirs = _setup_cpu()
    names = {n for n, _ in enc.named_parameters()}
    stored = {k[len('grad.enc.'):].rsplit('.', 1)[0] for k in z.files if k.startswith('grad.enc.')}
    assert names == stored                      # one reference gradient per parameter of ours, by name
    assert 0.15 < float(z['visible_frac.hw']) < 0.25      # most pillar points leave the images: compaction paths run
