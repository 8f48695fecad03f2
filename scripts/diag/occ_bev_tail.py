"""Where do the 6 ms of the nuscenes_occ_bev occupancy tail go (scripts/bench_hotpath_all.py: resample_lut_iou_counts 6.0 ms
against 0.18 ms for nuscenes_occ)?  Host timers with synchronisation around each call + GPU events."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hotpath_common as hc
from selfocc_amd.occ import occ_resample, MeanIoU, OPENSEED2NUSCENES
os.environ['eval'] = 'true'
d = torch.device('cuda:0')
for name in sys.argv[1:] or ['nuscenes_occ_bev', 'nuscenes_occ']:
    cfg = hc.shipped_for_eval(name)
    mods = hc.build(cfg, d)
    for m in mods[:3]:
        m.eval()
    fr = hc.frame_inputs(cfg, name, d, seed=1, want_images=False)
    state = {}
    with torch.no_grad():
        for it in range(3):
            hc.eval_entry(mods, cfg, name, fr, state)
        torch.cuda.synchronize()
        lifter, encoder, head = mods[:3]
        rep = encoder(lifter(fr[1])['representation'], ms_img_feats=fr[1], metas=fr[0])['representation']
        res = head.forward_occ(rep, fr[0], aabb=cfg['model']['head']['roi_aabb'], resolution=0.4)
        torch.cuda.synchronize()
        print(name, 'sdf', tuple(res['sdf'].shape), res['sdf'].is_contiguous(), 'logits', tuple(res['logits'].shape), res['logits'].is_contiguous(),
              res['logits'].dtype, 'sdf range', float(res['sdf'].min()), float(res['sdf'].max()), 'occupied', float((res['sdf'] <= 0).float().mean()))
        g = hc._OCC3D_GRID[str(d)]
        for rep_i in range(3):
            t0 = time.perf_counter(); e0 = hc.ev()
            got = occ_resample(res['sdf'], g, 0.0, logits=res['logits'], lut=OPENSEED2NUSCENES, crop=(6, 6, 6, 6, 0, 4))
            e1 = hc.ev(); torch.cuda.synchronize(); t1 = time.perf_counter()
            state['miou']._after_step(got['sem'], state['gt'], state['mask'])
            e2 = hc.ev(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print(json.dumps(dict(resample_host_ms=round((t1 - t0) * 1e3, 3), resample_gpu_ms=round(e0.elapsed_time(e1), 3),
                                  iou_host_ms=round((t2 - t1) * 1e3, 3), iou_gpu_ms=round(e1.elapsed_time(e2), 3),
                                  occ_frac=float(got['occ'].float().mean()))))
    del mods
    torch.cuda.empty_cache()
