"""ALL SEVEN shipped experiment configs (config/{nuscenes,kitti,kitti_raw}/*.py -> scripts/shipped_cfg/*.json), each built through
the registries at its shipped shapes, timed in ONE process: a training iteration (train.py:219-239: encoder / head / losses
forward, backward; clip_grad_norm_ + AdamW beside it) and the evaluation entry the docs pair with the config with the reference's
eval-time overrides (scripts/hotpath_common.py: SHIPPED).  Synthetic stand-ins for what is out of scope (camera rig, FPN maps,
images).  Prints one JSON line: {config: {train: {...stage ms, total}, eval: {...stage ms, total}}}.
Stage times are the MEDIAN over the timed iterations, with Python's cyclic GC collected before and held off during them: a 3 ms
eval frame whose enqueue takes 1.5 ms of host time shows a single 30 ms collection (the training modules of the same config
were just deleted) as + 6 ms on the mean of five frames — seen twice in round 6, `host_enqueue_ms` 6 - 8 instead of 1.5 - 2.3.
    python scripts/bench_hotpath_all.py [--only NAME[,NAME]] [--iters 5]"""
import argparse
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import hotpath_common as hc

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--warm", type=int, default=2)
ap.add_argument("--no-train", action="store_true")
ap.add_argument("--no-eval", action="store_true")
args = ap.parse_args()
d = torch.device("cuda:0")
names = [n for n in hc.SHIPPED if not args.only or n in args.only.split(",")]
res = {}


def mean_stages(evs, keys):
    """median over the timed iterations of every stage (see the module docstring)"""
    out = {}
    for k, (a, b) in keys.items():
        out[k] = round(float(np.median([e[a].elapsed_time(e[b]) for e in evs])), 3)
    return out


for name in names:
    torch.manual_seed(0); np.random.seed(0)
    r = {}
    if not args.no_train:
        os.environ['eval'] = 'false'
        cfg = hc.shipped(name)
        mods = hc.build(cfg, d, want_loss=True)
        for m in mods[:3]:
            m.train()
        params = [p for m in mods[:3] for p in m.parameters()]
        opt_cfg = dict(cfg['optimizer']['optimizer'])
        assert opt_cfg.pop('type') == 'AdamW'
        optimizer = torch.optim.AdamW(params, **opt_cfg)
        fr = hc.frame_inputs(cfg, name, d, seed=0)
        evs = []
        host = []
        gc.collect(); gc.disable()
        for it in range(args.warm + args.iters):
            optimizer.zero_grad(set_to_none=True)
            e = {}
            h0 = time.perf_counter()
            hc.train_iteration(mods, cfg, fr, global_iter=it, events=e)
            h1 = time.perf_counter()
            torch.nn.utils.clip_grad_norm_(params, cfg['grad_max_norm'])
            optimizer.step()
            e['t5'] = hc.ev()
            torch.cuda.synchronize()
            if it >= args.warm:
                evs.append(e)
                host.append((h1 - h0) * 1e3)
        gc.enable()
        t = mean_stages(evs, dict(encoder_fwd=('t0', 't1'), head_fwd=('t1', 't2'), losses_fwd=('t2', 't3'), backward=('t3', 't4'),
                                  clip_and_adamw=('t4', 't5')))
        t['total_ms'] = round(sum(v for k, v in t.items() if k != 'clip_and_adamw'), 2)
        t['host_enqueue_ms'] = round(float(np.median(host)), 2)     # the python / launch side of the same iteration (no device wait)
        t['rays'] = cfg['num_rays'][0] * cfg['num_rays'][1] * cfg['model']['encoder']['num_cams']
        t['losses'] = [c['type'] for c in cfg['loss']['loss_cfgs']]
        r['train'] = t
        del mods, params, optimizer, fr, evs
        torch.cuda.empty_cache()
    if not args.no_eval:
        os.environ['eval'] = 'true'
        cfg = hc.shipped_for_eval(name)
        mods = hc.build(cfg, d)
        for m in mods[:3]:
            m.eval()
        fr = hc.frame_inputs(cfg, name, d, seed=1, want_images=False)
        evs, state, host = [], {}, []
        gc.collect(); gc.disable()
        with torch.no_grad():
            for it in range(args.warm + args.iters):
                e = {}
                h0 = time.perf_counter()
                out = hc.eval_entry(mods, cfg, name, fr, state, events=e)
                h1 = time.perf_counter()
                torch.cuda.synchronize()
                if it >= args.warm:
                    evs.append(e)
                    host.append((h1 - h0) * 1e3)
        gc.enable()
        kind = hc.SHIPPED[name]['eval']
        second = {'render': 'prepare_volume', 'render_novel': 'prepare_volume', 'occ3d': 'volume_and_dense_query',
                  'occ_kitti': 'volume_and_dense_query'}[kind]
        third = {'render': 'render', 'render_novel': 'render', 'occ3d': 'resample_lut_iou_counts', 'occ_kitti': 'threshold_crop_iou_counts'}[kind]
        t = mean_stages(evs, {'encoder_fwd': ('t0', 't1'), second: ('t1', 't2'), third: ('t2', 't3')})
        t['total_ms'] = round(sum(t.values()), 2)
        t['host_enqueue_ms'] = round(float(np.median(host)), 2)
        t['entry'] = kind
        if kind.startswith('render'):
            t['rays'] = int(out['ms_depths'][0].numel())
            t['depth_mean'] = round(float(out['ms_depths'][0].mean()), 3)
        else:
            t['occupied_frac'] = round(float(out['occ'].float().mean()), 4)
        r['eval'] = t
        os.environ['eval'] = 'false'
        del mods, fr, evs, state, out
        torch.cuda.empty_cache()
    r['built_from'] = hc.shipped(name)['source']
    res[name] = r
res['stat'] = f'median of {args.iters} iterations after {args.warm} warm-up, GC held off'
res['max_mem_GB'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
print(json.dumps(res))
