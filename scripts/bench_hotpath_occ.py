"""Stage timing of ONE Occ3D-nuScenes occupancy-evaluation frame of the hot path at the shipped nuscenes_occ shapes
(eval_iou.py:150-260): TPV 257x257x25, aabb +-40 x [-1, 5.4], 25-channel volume (sdf + rgb + 21 OpenSeeD logits);
encoder -> NeuSHead.forward_occ (fused tri-plane MLP + dense 200x200x16 query with logits) -> ego-frame resample /
threshold / crop / arg-max / LUT (selfocc_occ_resample) -> integer IoU counts (MeanIoU).  Random FPN features stand in
for ResNet50 + FPN and random voxel labels for the Occ3D ground truth.  JSON of per-stage milliseconds."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import hotpath_common as hc
from selfocc_amd.occ import occ_resample, MeanIoU, OPENSEED2NUSCENES

os.environ['eval'] = 'true'
d = torch.device("cuda:0")
torch.manual_seed(0)
cfg = hc.modify_for_eval(hc.shipped("nuscenes_occ"), 'nuscenes')          # built from the SHIPPED config/nuscenes/nuscenes_occ.py
lifter, encoder, head, _ = hc.build(cfg, d)
encoder.eval(); head.eval(); lifter.eval()
img = tuple(cfg['img_size'])
pcr = cfg['model']['head']['roi_aabb']
c2w, l2i, K = hc.ring_cameras(6, img, 1266.0)
metas = [dict(lidar2img=l2i, img2lidar=c2w, img_shape=img)]
feats = hc.fpn_feats(6, cfg['model']['encoder']['embed_dims'], img, d)
# ego -> lidar resampling coordinates of the Occ3D grid (normalised to [-1, 1] along (H<->y, W<->x, D<->z)), eval_iou.py:211-232
g = torch.stack(torch.meshgrid(torch.linspace(0.015, 0.985, 200), torch.linspace(0.02, 0.99, 200), torch.linspace(0.05, 0.95, 16),
                               indexing='ij'), -1).to(d).contiguous()          # normalised [0, 1] along (H, W, D)
gt = gt_mask = None
cls = list(range(1, 17))
miou = MeanIoU(cls, 0, [str(c) for c in cls], True, 0)
miou.reset()
ev = hc.ev
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
out['built_from'] = cfg['source']
out['max_mem_GB'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
print(json.dumps(out))
