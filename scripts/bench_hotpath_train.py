"""Stage timing of ONE training iteration of the hot path, built from the SHIPPED config/nuscenes/nuscenes_occ.py
(scripts/shipped_cfg/nuscenes_occ.json -> registries: TPVQueryLifter, TPVFormerEncoder x 4 layers, NeuSHead, MultiLoss with the
five shipped losses wired by the shipped loss_input_convertion; train.py:219-242): TPV 257x257x25, 6 cameras, 48x100 cellular
rays, 256 samples, color_dims 24.  Synthetic inputs (random FPN features instead of ResNet50 + FPN, which is out of scope).
Stages: encoder / head / losses forward, backward — and, reported beside them, the optimiser side of train.py:239-242
(clip_grad_norm_(grad_max_norm) + AdamW.step over the hot path's parameters).  `iteration_total_ms` = forward + backward (the
figure of every earlier round); `iteration_with_optimizer_ms` adds clip + step.  JSON of per-stage milliseconds."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import hotpath_common as hc
from selfocc_amd.model import bricks as _bricks
_bricks.HEAD_MAJOR_VALUE = os.environ.get('SO_HEAD_MAJOR', '0') == '1'   # A/B switch of the MSDA value layout

d = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(0)
cfg = hc.shipped("nuscenes_occ")
lifter, encoder, head, loss_fn = hc.build(cfg, d, want_loss=True)
img = tuple(cfg['img_size'])
dim = cfg['model']['encoder']['embed_dims']
c2w, l2i, K = hc.ring_cameras(6, img, 1266.0)
metas = [dict(lidar2img=l2i, img2lidar=c2w, temImg2lidar=c2w, img_shape=img,
              img2prevImg=np.stack([hc.motion(K, 2, 0.3, -0.8)] * 6), img2nextImg=np.stack([hc.motion(K, -2, -0.3, 0.8)] * 6),
              sem=torch.randint(0, 21, (6, *img), device=d))]
feats = hc.fpn_feats(6, dim, img, d)
imgs = {k: torch.rand(1, 6, 3, *img, device=d) for k in ('curr_imgs', 'prev_imgs', 'next_imgs', 'color_imgs')}
params = list(lifter.parameters()) + list(encoder.parameters()) + list(head.parameters())
opt_cfg = dict(cfg['optimizer']['optimizer'])
assert opt_cfg.pop('type') == 'AdamW'
optimizer = torch.optim.AdamW(params, **opt_cfg)
stages = {}


def run(n_iter, record):
    for it in range(n_iter):
        optimizer.zero_grad(set_to_none=True)
        e0 = hc.ev()
        rep = lifter(feats)['representation']
        rep = encoder(rep, ms_img_feats=feats, metas=metas)['representation']
        e1 = hc.ev()
        out = head(rep, metas, global_iter=it)
        e2 = hc.ev()
        loss_input = dict(metas=metas, curr_feats=imgs['curr_imgs'], prev_feats=imgs['prev_imgs'], next_feats=imgs['next_imgs'], **imgs)
        for k, v in cfg['loss_input_convertion'].items():
            loss_input[k] = out[v]
        total, parts = loss_fn(loss_input)
        e3 = hc.ev()
        total.backward()
        e4 = hc.ev()
        torch.nn.utils.clip_grad_norm_(params, cfg['grad_max_norm'])
        optimizer.step()
        e5 = hc.ev()
        torch.cuda.synchronize()
        if record:
            for k, (a, b) in dict(encoder_fwd=(e0, e1), head_fwd=(e1, e2), losses_fwd=(e2, e3), backward_all=(e3, e4),
                                  clip_and_adamw=(e4, e5)).items():
                stages.setdefault(k, []).append(a.elapsed_time(b))


encoder.train(); head.train()
run(2, False)
run(5, True)
res = {k: round(sum(v) / len(v), 2) for k, v in stages.items()}
res['iteration_total_ms'] = round(sum(v for k, v in res.items() if k != 'clip_and_adamw'), 2)
res['iteration_with_optimizer_ms'] = round(res['iteration_total_ms'] + res['clip_and_adamw'], 2)
res['built_from'] = cfg['source']
res['max_mem_GB'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
print(json.dumps(res))
