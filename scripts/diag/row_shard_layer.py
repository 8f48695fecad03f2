"""Diagnostic (GPU, one process): one encoder layer on the LOCAL rows of a 2-way row shard (exactly the call
TPVFormerEncoder._forward_layers_sharded makes) against the same rows of the unsharded layer, at the shipped size, with the
fast paths toggled."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import hotpath_common as hc
from selfocc_amd.dist import PlaneRowShard
from selfocc_amd.model import bricks
from selfocc_amd.model.encoder import tpvformer as tf
from selfocc_amd.model.encoder.utils import point_sampling

d = torch.device("cuda:0")
torch.manual_seed(0)
cfg = hc.shipped("nuscenes_occ")
lifter, enc, _h, _ = hc.build(cfg, d)
enc.eval()
img = tuple(cfg['img_size'])
c2w, l2i, K = hc.ring_cameras(6, img, 1266.0)
metas = [dict(lidar2img=l2i, img2lidar=c2w, img_shape=img)]
g = torch.Generator().manual_seed(5)
feats = [torch.randn(1, 6, 96, -(-img[0] // s_), -(-img[1] // s_), generator=g).to(d) for s_ in (8, 16, 32, 64)]
H, W, Z = enc.tpv_size
sizes = [H * W, Z * H, W * Z]
with torch.no_grad():
    rep = lifter(feats)['representation']
    tpv_pos, tpv_pos_cat = enc._positions(1)
    feat, spatial_shapes, level_start_index = enc._flatten_feats(feats)
    cams, masks = [], []
    for ref_3d in (enc.ref_3d_hw, enc.ref_3d_zh, enc.ref_3d_wz):
        c, m = point_sampling(ref_3d.unsqueeze(0), metas)
        cams.append(c); masks.append(m)
    ref_cv = enc.cross_view_ref_points.unsqueeze(0)
    if tpv_pos_cat is None:
        tpv_pos_cat = torch.cat(tpv_pos, dim=1)
    # emulate the two ranks in ONE process, all layers: each "rank" computes its local rows from the (emulated) gathered planes
    q_ref = rep
    q_sh = tf._as_cat(rep)
    for li, layer in enumerate(enc.layers):
        full = layer(q_ref, feat, feat, tpv_pos=tpv_pos, tpv_pos_cat=tpv_pos_cat, ref_2d=ref_cv, spatial_shapes=spatial_shapes,
                     level_start_index=level_start_index, reference_points_cams=cams, tpv_masks=masks, tpv_size=enc.tpv_size,
                     rebatch_plans=None)
        q_ref = full
        full = tf._as_cat(full)
        nxt = torch.empty_like(q_sh)
        for rank in (0, 1):
            shard = PlaneRowShard(sizes, rank, 2)
            pos_loc = shard.take(tpv_pos_cat, 1)
            call = dict(tpv_pos=list(torch.split(pos_loc, shard.local_sizes, 1)), tpv_pos_cat=pos_loc, ref_2d=shard.take(ref_cv, 1),
                        spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                        reference_points_cams=[shard.take_plane(c, i, 2) for i, c in enumerate(cams)],
                        tpv_masks=[shard.take_plane(m, i, 2) for i, m in enumerate(masks)], tpv_size=enc.tpv_size,
                        rebatch_plans=None, plane_sizes=shard.local_sizes, self_attn_value=q_sh)
            out = tf._as_cat(layer(tf._as_planes(shard.take(q_sh, 1), shard.local_sizes), feat, feat, **call))
            off_g, off_l = 0, 0
            for n, (a, b) in zip(sizes, shard.local):
                nxt[:, off_g + a:off_g + b] = out[:, off_l:off_l + (b - a)]
                off_g += n; off_l += b - a
        errs = [round(((a - b).abs().max() / b.abs().max()).item(), 7) for a, b in zip(torch.split(nxt, sizes, 1), torch.split(full, sizes, 1))]
        print(f"layer {li}: emulated two-rank chain vs unsharded chain, plane errors {errs}", flush=True)
        q_sh = nxt
