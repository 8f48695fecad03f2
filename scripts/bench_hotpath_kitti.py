"""Stage timing of ONE SemanticKITTI novel-depth evaluation frame (BASELINE configs[3]; eval_novel_depth.py), built from the
SHIPPED config/kitti/kitti_novel_depth.py (scripts/shipped_cfg/kitti_novel_depth.json) with the reference's eval-time overrides
(utils/config_tools.py: 176x608 fixed lattice, trans_kw render_img2lidar = the novel view's matrix): one camera, image 370x1216,
TPV 257x257x33, sdf + rgb volume, 256 samples.  Random FPN features stand in for the backbone.  JSON of per-stage milliseconds."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import hotpath_common as hc

os.environ['eval'] = 'true'
d = torch.device("cuda:0")
torch.manual_seed(0)
cfg = hc.modify_for_eval(hc.shipped("kitti_novel_depth"), 'kitti', novel_depth=True)
lifter, encoder, head, _ = hc.build(cfg, d)
encoder.eval(); head.eval(); lifter.eval()
img = tuple(cfg['img_size'])
dim = cfg['model']['encoder']['embed_dims']
n_cams = cfg['model']['encoder']['num_cams']
c2w, l2i, K = hc.ring_cameras(n_cams, img, 707.0, z=1.7)
novel = c2w.copy()
novel[:, 1, 3] += 1.0                                   # the rendered view: one metre further along the road
metas = [dict(lidar2img=l2i, img2lidar=c2w, render_img2lidar=novel, img_shape=img)]
feats = hc.fpn_feats(n_cams, dim, img, d)
st = {}
with torch.no_grad():
    for it in range(7):
        e0 = hc.ev()
        rep = encoder(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        e1 = hc.ev()
        head.prepare(rep, metas)
        e2 = hc.ev()
        out = head.render(metas, batch=90000)
        e3 = hc.ev()
        torch.cuda.synchronize()
        if it >= 2:
            for k, (a, b) in dict(encoder_fwd=(e0, e1), prepare_volume=(e1, e2), render_107k_rays=(e2, e3)).items():
                st.setdefault(k, []).append(a.elapsed_time(b))
res = {k: round(sum(v) / len(v), 3) for k, v in st.items()}
res['frame_total_ms'] = round(sum(res.values()), 2)
res['depth_mean'] = round(out['ms_depths'][0].mean().item(), 3)
res['n_rays'] = int(out['ms_depths'][0].numel())
res['built_from'] = cfg['source']
res['max_mem_GB'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
print(json.dumps(res))
