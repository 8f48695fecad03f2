"""Stage timing of ONE training iteration of the hot path at the shipped nuscenes_occ shapes
(SURVEY Appendix C): TPV 257x257x25, C=96, 6 heads, 4 encoder layers, FPN maps 96x200 / 48x100 /
24x50 / 12x25 x 6 cams, 48x100 cellular rays x 6 cams, 256 samples, color_dims 24, all five
losses.  Synthetic inputs (random FPN features instead of ResNet50+FPN, which is out of scope).
Prints a JSON dict of per-stage milliseconds (HIP events, mean over iterations)."""
import sys, os, json, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selfocc_amd.registry import MODELS, OPENOCC_LOSS
import selfocc_amd.model, selfocc_amd.loss  # noqa
from selfocc_amd.model import bricks as _bricks
_bricks.HEAD_MAJOR_VALUE = os.environ.get('SO_HEAD_MAJOR', '0') == '1'   # A/B switch of the MSDA value layout

d = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(0)
dim, heads = 96, 6
mapping_args = dict(nonlinear_mode='linear', h_size=[128, 0], h_range=[40.0, 0], h_half=False, w_size=[128, 0],
                    w_range=[40.0, 0], w_half=False, d_size=[24, 0], d_range=[-1.0, 5.4, 5.4])
pcr = [-40.0, -40.0, -1.0, 40.0, 40.0, 5.4]
H = W = 257; Z = 25
layer = dict(type='TPVFormerLayer',
             attn_cfgs=[dict(type='CrossViewHybridAttention', embed_dims=dim, num_heads=heads, num_levels=3, num_points=12, dropout=0.1, batch_first=True),
                        dict(type='TPVCrossAttention', embed_dims=dim, num_cams=6, dropout=0.1, batch_first=True, num_heads=heads, num_levels=4, num_points=[48, 48, 8])],
             feedforward_channels=2 * dim, ffn_dropout=0.1, operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'))
lifter = MODELS.build(dict(type='TPVQueryLifter', tpv_h=H, tpv_w=W, tpv_z=Z, dim=dim)).to(d)
encoder = MODELS.build(dict(type='TPVFormerEncoder', mapping_args=mapping_args, embed_dims=dim, num_cams=6, num_feature_levels=4,
                            positional_encoding=dict(type='TPVPositionalEncoding', num_freqs=[12] * 3, embed_dims=dim, tot_range=pcr),
                            num_points_cross=[48, 48, 8], num_points_self=[12] * 3, transformerlayers=[layer] * 4, num_layers=4)).to(d)
encoder.init_weights()
img_size, num_rays = [768, 1600], [48, 100]
head = MODELS.build(dict(type='NeuSHead', roi_aabb=pcr, resolution=0.4, num_samples=256, num_samples_importance=0, num_up_sample_steps=0,
                         beta_init=0.2, use_numerical_gradients=False, sample_gradient=True, return_second_grad=True, return_sem=True,
                         ray_sample_mode='cellular', ray_number=num_rays, ray_img_size=img_size, trans_kw='temImg2lidar',
                         render_bkgd='random', mapping_args=mapping_args, embed_dims=dim, color_dims=24, density_layers=2, sh_deg=0,
                         two_split=False, tpv=True)).to(d)
keys = {'curr_imgs': 'curr_imgs', 'prev_imgs': 'prev_imgs', 'next_imgs': 'next_imgs', 'ray_indices': 'ray_indices',
        'weights': 'weights', 'ts': 'ts', 'metas': 'metas', 'ms_rays': 'ms_rays'}
loss_fn = OPENOCC_LOSS.build(dict(type='MultiLoss', sync_items=False, loss_cfgs=[
    dict(type='ReprojLossMonoMultiNewCombine', weight=1.0, no_ssim=False, img_size=img_size, ray_resize=num_rays, input_dict=keys),
    dict(type='RGBLossMS', weight=0.1, img_size=img_size, no_ssim=False, ray_resize=num_rays,
         input_dict={'ms_colors': 'ms_colors', 'ms_rays': 'ms_rays', 'gt_imgs': 'curr_imgs'}),
    dict(type='EikonalLoss', weight=0.1), dict(type='SecondGradLoss', weight=0.01),
    dict(type='SemCELossMS', weight=0.1, img_size=img_size, ray_resize=num_rays)]))

# cameras: 6 pinholes at the ego origin
K = np.array([[1266.0, 0, 800, 0], [0, 1266.0, 384, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
c2w, l2i = [], []
for i in range(6):
    yaw = math.radians(60 * i)
    fwd = np.array([math.cos(yaw), math.sin(yaw), 0.0]); right = np.array([math.sin(yaw), -math.cos(yaw), 0.0]); down = np.array([0, 0, -1.0])
    m = np.eye(4); m[:3, :3] = np.stack([right, down, fwd], 1); m[:3, 3] = [0.2 * i, 0.1, 1.5]
    c2w.append(m @ np.linalg.inv(K)); l2i.append(K @ np.linalg.inv(m))
def motion(yaw, tx, tz):
    y = np.deg2rad(yaw); Rm = np.array([[np.cos(y), 0, np.sin(y), tx], [0, 1, 0, 0], [-np.sin(y), 0, np.cos(y), tz], [0, 0, 0, 1]])
    return K @ Rm @ np.linalg.inv(K)
metas = [dict(lidar2img=np.stack(l2i), img2lidar=np.stack(c2w), temImg2lidar=np.stack(c2w), img_shape=(768, 1600),
              img2prevImg=np.stack([motion(2, 0.3, -0.8)] * 6), img2nextImg=np.stack([motion(-2, -0.3, 0.8)] * 6),
              sem=torch.randint(0, 21, (6, 768, 1600), device=d))]
feats = [torch.randn(1, 6, dim, h, w, device=d) for h, w in ((96, 200), (48, 100), (24, 50), (12, 25))]
imgs = {k: torch.rand(1, 6, 3, 768, 1600, device=d) for k in ('curr_imgs', 'prev_imgs', 'next_imgs')}
params = list(lifter.parameters()) + list(encoder.parameters()) + list(head.parameters())

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
stages = {}
def run(n_iter, record):
    for it in range(n_iter):
        for p in params: p.grad = None
        e0 = ev()
        rep = lifter(feats)['representation']
        rep = encoder(rep, ms_img_feats=feats, metas=metas)['representation']
        e1 = ev()
        out = head(rep, metas, global_iter=it)
        e2 = ev()
        total, parts = loss_fn(dict(out, metas=metas, **imgs))
        e3 = ev()
        total.backward()
        e4 = ev()
        torch.cuda.synchronize()
        if record:
            for k, (a, b) in dict(encoder_fwd=(e0, e1), head_fwd=(e1, e2), losses_fwd=(e2, e3), backward_all=(e3, e4)).items():
                stages.setdefault(k, []).append(a.elapsed_time(b))
encoder.train(); head.train()
run(2, False)
run(5, True)
res = {k: round(sum(v) / len(v), 2) for k, v in stages.items()}
res['iteration_total_ms'] = round(sum(res.values()), 2)
res['max_mem_GB'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
print(json.dumps(res))
