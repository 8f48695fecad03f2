"""Stage timing of ONE full-resolution depth-evaluation frame of the hot path at the shipped
nuscenes_depth shapes (eval_depth.py:150-227, the path behind the reference's "about 90 min"):
TPV 257x257x31, aabb +-51.2 x [-4, 5], color_dims 0, 6 x 450x800 rays, 256 samples.  Random FPN
features stand in for ResNet50+FPN (out of scope).  JSON of per-stage milliseconds."""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selfocc_amd.registry import MODELS
import selfocc_amd.model  # noqa
from selfocc_amd.model import bricks as _bricks
_bricks.HEAD_MAJOR_VALUE = os.environ.get('SO_HEAD_MAJOR', '0') == '1'   # A/B switch of the MSDA value layout

os.environ['eval'] = 'true'
d = torch.device("cuda:0")
torch.manual_seed(0)
dim, heads = 96, 6
mapping_args = dict(nonlinear_mode='linear', h_size=[128, 0], h_range=[51.2, 0], h_half=False, w_size=[128, 0],
                    w_range=[51.2, 0], w_half=False, d_size=[30, 0], d_range=[-4.0, 5.0, 5.0])
pcr = [-51.2, -51.2, -4.0, 51.2, 51.2, 5.0]
H = W = 257; Z = 31
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
                         beta_init=0.3, use_numerical_gradients=False, sample_gradient=True, return_max_depth=True,
                         ray_sample_mode='fixed', ray_number=[450, 800], ray_img_size=[900, 1600], trans_kw='img2lidar',
                         render_bkgd='random', mapping_args=mapping_args, embed_dims=dim, color_dims=0, density_layers=2, sh_deg=0,
                         two_split=False, tpv=True)).to(d)
encoder.eval(); head.eval(); lifter.eval()
K = np.array([[1266.0, 0, 800, 0], [0, 1266.0, 450, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
c2w, l2i = [], []
for i in range(6):
    yaw = math.radians(60 * i)
    fwd = np.array([math.cos(yaw), math.sin(yaw), 0.0]); right = np.array([math.sin(yaw), -math.cos(yaw), 0.0]); down = np.array([0, 0, -1.0])
    m = np.eye(4); m[:3, :3] = np.stack([right, down, fwd], 1); m[:3, 3] = [0.2 * i, 0.1, 1.5]
    c2w.append(m @ np.linalg.inv(K)); l2i.append(K @ np.linalg.inv(m))
metas = [dict(lidar2img=np.stack(l2i), img2lidar=np.stack(c2w), img_shape=(900, 1600))]
feats = [torch.randn(1, 6, dim, h, w, device=d) for h, w in ((112, 200), (56, 100), (28, 50), (14, 25))]

import selfocc_amd.model.encoder.tpvformer as T
import selfocc_amd.model.encoder.utils as U
times = {}
def timed(name, fn):
    def w(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        times.setdefault(name, []).append((e0, e1))
        return r
    return w
T.point_sampling = timed('point_sampling', T.point_sampling)
encoder.positional_encoding.forward = timed('pos_enc', encoder.positional_encoding.forward)
encoder._flatten_feats = timed('flatten_feats', encoder._flatten_feats)
for li, layer in enumerate(encoder.layers):
    for ai, att in enumerate(layer.attentions):
        att.forward = timed('self_attn' if ai == 0 else 'cross_attn', att.forward)
    for n in layer.norms: n.forward = timed('norm', n.forward)
    for f in layer.ffns: f.forward = timed('ffn', f.forward)
    layer.forward = timed('layer_total', layer.forward)
encoder.forward_layers = timed('forward_layers', encoder.forward_layers)
with torch.no_grad():
    for it in range(6):
        if it == 2: times.clear()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        rep = encoder(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        e1.record(); times.setdefault('encoder_total', []).append((e0, e1))
torch.cuda.synchronize()
out = {k: round(sum(a.elapsed_time(b) for a, b in v) / 4, 3) for k, v in times.items()}
print(json.dumps(out))
