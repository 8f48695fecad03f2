"""Stage timing of ONE Occ3D-nuScenes occupancy-evaluation frame of the hot path at the shipped nuscenes_occ shapes
(eval_iou.py:150-260): TPV 257x257x25, aabb +-40 x [-1, 5.4], 25-channel volume (sdf + rgb + 21 OpenSeeD logits);
encoder -> NeuSHead.forward_occ (fused tri-plane MLP + dense 200x200x16 query with logits) -> ego-frame resample /
threshold / crop / arg-max / LUT (selfocc_occ_resample) -> integer IoU counts (MeanIoU).  Random FPN features stand in
for ResNet50 + FPN and random voxel labels for the Occ3D ground truth.  JSON of per-stage milliseconds."""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selfocc_amd.registry import MODELS
import selfocc_amd.model  # noqa
from selfocc_amd.occ import occ_resample, MeanIoU, OPENSEED2NUSCENES

os.environ['eval'] = 'true'
d = torch.device("cuda:0")
torch.manual_seed(0)
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
head = MODELS.build(dict(type='NeuSHead', roi_aabb=pcr, resolution=0.4, num_samples=256, num_samples_importance=0, num_up_sample_steps=0,
                         beta_init=0.3, use_numerical_gradients=False, sample_gradient=True, return_sem=True,
                         ray_sample_mode='fixed', ray_number=[48, 100], ray_img_size=[768, 1600], trans_kw='img2lidar',
                         render_bkgd='random', mapping_args=mapping_args, embed_dims=dim, color_dims=24, density_layers=2, sh_deg=0,
                         two_split=False, tpv=True)).to(d)
encoder.eval(); head.eval(); lifter.eval()
K = np.array([[1266.0, 0, 800, 0], [0, 1266.0, 384, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
c2w, l2i = [], []
for i in range(6):
    yaw = math.radians(60 * i)
    fwd = np.array([math.cos(yaw), math.sin(yaw), 0.0]); right = np.array([math.sin(yaw), -math.cos(yaw), 0.0]); down = np.array([0, 0, -1.0])
    m = np.eye(4); m[:3, :3] = np.stack([right, down, fwd], 1); m[:3, 3] = [0.2 * i, 0.1, 1.5]
    c2w.append(m @ np.linalg.inv(K)); l2i.append(K @ np.linalg.inv(m))
metas = [dict(lidar2img=np.stack(l2i), img2lidar=np.stack(c2w), img_shape=(768, 1600))]
feats = [torch.randn(1, 6, dim, h, w, device=d) for h, w in ((96, 200), (48, 100), (24, 50), (12, 25))]
# ego -> lidar resampling coordinates of the Occ3D grid (normalised to [-1, 1] along (H<->y, W<->x, D<->z)), eval_iou.py:211-232
g = torch.stack(torch.meshgrid(torch.linspace(0.015, 0.985, 200), torch.linspace(0.02, 0.99, 200), torch.linspace(0.05, 0.95, 16),
                               indexing='ij'), -1).to(d).contiguous()          # normalised [0, 1] along (H, W, D)
gt = gt_mask = None
cls = list(range(1, 17))
miou = MeanIoU(cls, 0, [str(c) for c in cls], True, 0)
miou.reset()
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
st = {}
with torch.no_grad():
    for it in range(7):
        e0 = ev()
        rep = encoder(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        e1 = ev()
        res = head.forward_occ(rep, metas, aabb=pcr, resolution=0.4)
        e2 = ev()
        got = occ_resample(res['sdf'], g, 0.0, logits=res['logits'], lut=OPENSEED2NUSCENES, crop=(6, 6, 6, 6, 0, 4))
        if gt is None:
            gt = torch.randint(0, 18, tuple(got['sem'].shape), device=d, dtype=torch.int32)
            gt_mask = torch.rand(tuple(got['sem'].shape), device=d) > 0.3
        miou._after_step(got['sem'], gt, gt_mask)
        e3 = ev()
        torch.cuda.synchronize()
        if it >= 2:
            for k, (a, b) in dict(encoder_fwd=(e0, e1), volume_and_dense_query=(e1, e2), resample_lut_iou_counts=(e2, e3)).items():
                st.setdefault(k, []).append(a.elapsed_time(b))
out = {k: round(sum(v) / len(v), 3) for k, v in st.items()}
out['frame_total_ms'] = round(sum(out.values()), 2)
out['occupied_frac'] = round(got['occ'].float().mean().item(), 4)
out['max_mem_GB'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
print(json.dumps(out))
