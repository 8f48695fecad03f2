"""GPU (one MI355X, two processes sharing cuda:0 over gloo — RCCL refuses two ranks on one device, so the
collectives are staged through the host here; production runs one rank per GPU over RCCL): the ray-sharded
NeuSHead == the unsharded NeuSHead, eval render and training gradients, with the real HIP kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, ws, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import test_head_gpu as th
        from selfocc_amd.registry import OPENOCC_LOSS
        d = th.D0
        msgs = []

        def build(shard):
            h = th.make_head(color_dims=8, return_sem=True, ray_sample_mode='fixed', ray_number=[7, 10],
                             render_bkgd='white', ray_shard=shard, single_jitter=True)
            return h

        # ---- eval: prepare + render, sharded vs not ----------------------------------------------------
        os.environ['eval'] = 'true'
        rep, metas, imgs = th.make_inputs()
        outs = []
        for shard in (False, True):
            h = build(shard).eval()
            with torch.no_grad():
                h.prepare(rep, metas)
                outs.append(h.render(metas))
        for k in ('ms_depths', 'ms_accs', 'ms_colors', 'ms_max_depths', 'sem'):
            a, b = outs[0][k][0], outs[1][k][0]
            # the shard's lattice offset oy + r0 * sy rounds differently from (iy + r0) * sy + oy: 1e-4, not bit-equal
            if not torch.allclose(a, b, rtol=1e-4, atol=1e-5):
                msgs.append(f"eval {k}: max diff {(a - b).abs().max().item():.3e}")
        # ---- train: forward + loss + backward; gradient of the planes and of the field parameters ----
        os.environ['eval'] = 'false'
        grads = []
        for shard in (False, True):
            h = build(shard).eval()      # eval(): no jitter / random background, so both runs see the same rays
            rep, metas, imgs = th.make_inputs()
            out = h(rep, metas, global_iter=0)
            loss_fn = OPENOCC_LOSS.build(dict(type='MultiLoss', sync_items=False, loss_cfgs=[
                dict(type='ReprojLossMonoMultiNewCombine', weight=1.0, no_ssim=False, img_size=[64, 64], ray_resize=[7, 10],
                     input_dict={'curr_imgs': 'curr_imgs', 'prev_imgs': 'prev_imgs', 'next_imgs': 'next_imgs',
                                 'ray_indices': 'ray_indices', 'weights': 'weights', 'ts': 'ts', 'metas': 'metas',
                                 'ms_rays': 'ms_rays'}),
                dict(type='RGBLossMS', weight=0.1, img_size=[64, 64], no_ssim=False, ray_resize=[7, 10],
                     input_dict={'ms_colors': 'ms_colors', 'ms_rays': 'ms_rays', 'gt_imgs': 'curr_imgs'}),
                dict(type='ReprojLossMonoMultiNew', weight=0.5, no_ssim=False, img_size=[64, 64], ray_resize=[7, 10],
                     input_dict={'curr_imgs': 'curr_imgs', 'prev_imgs': 'prev_imgs', 'next_imgs': 'next_imgs',
                                 'ray_indices': 'ray_indices', 'weights': 'weights', 'ts': 'ts', 'metas': 'metas',
                                 'ms_rays': 'ms_rays', 'deltas': 'deltas'}),
                dict(type='EikonalLoss', weight=0.1)]))
            if shard:      # the per-sample tensors stayed on the rank that rendered them (no 206 MB all-gather)
                from selfocc_amd.dist import LocalRows, shard_of
                assert isinstance(out['weights'], LocalRows) and shard_of(out['eik_grad']) is not None
                assert out['weights'][0].numel() < 7 * 10 * 32 and out['ms_depths'][0].shape == (1, 2, 70)
            total, _ = loss_fn(dict(out, metas=metas, **imgs))
            total.backward()
            grads.append(([r.grad.clone() for r in rep],
                          {n: p.grad.clone() for n, p in h.named_parameters() if p.grad is not None}, total.detach()))
        if not torch.allclose(grads[0][2], grads[1][2], rtol=1e-4, atol=1e-6):
            msgs.append(f"loss {grads[0][2].item()} vs {grads[1][2].item()}")
        for i, (a, b) in enumerate(zip(grads[0][0], grads[1][0])):
            scale = a.abs().max().item()
            if (a - b).abs().max().item() > 1e-4 * scale + 1e-9:
                msgs.append(f"plane {i} grad: max diff {(a - b).abs().max().item():.3e} of {scale:.3e}")
        for n in grads[0][1]:
            a, b = grads[0][1][n], grads[1][1][n]
            scale = a.abs().max().item()
            if (a - b).abs().max().item() > 1e-4 * scale + 1e-9:
                msgs.append(f"param {n} grad: max diff {(a - b).abs().max().item():.3e} of {scale:.3e}")
        # ---- guard: ray sharding splits ONE frame; ranks that were fed different frames must all raise ----
        h = build(True).eval()
        rep, metas, imgs = th.make_inputs()
        if rank == 1:
            metas[0]['temImg2lidar'] = np.array(metas[0]['temImg2lidar']) + 0.5
        try:
            h(rep, metas, global_iter=0)
            msgs.append("different frames on the ranks did not raise")
        except RuntimeError as e:
            if "DIFFERENT frames" not in str(e):
                raise
        # ... also when the calibration is the SAME on every rank (KITTI, a static rig) and only the frame differs — on a
        # LATER call (round-4 advisor finding: the guard compared matrices only, and only every 64th call)
        h = build(True).eval()
        rep, metas, imgs = th.make_inputs()
        metas[0]['token'] = 'frame-0'
        h(rep, metas, global_iter=0)
        h(rep, metas, global_iter=1)
        metas[0]['token'] = 'frame-0' if rank == 0 else 'frame-7'
        try:
            h(rep, metas, global_iter=2)
            msgs.append("different frame tokens (same calibration) on the third call did not raise")
        except RuntimeError as e:
            if "DIFFERENT frames" not in str(e):
                raise
        ret[rank] = msgs
    except Exception as e:   # surface the failure in the parent
        import traceback
        ret[rank] = [f"exception: {e!r}\n{traceback.format_exc()}"]
    finally:
        dist.destroy_process_group()


def test_ray_sharded_head_equals_unsharded_world2(hip):
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for r in range(ws):
        assert ret.get(r) == [], f"rank {r}: {ret.get(r)}"


def _enc_worker(rank, ws, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import test_golden_encoder_full_gpu as tf
        msgs = []
        z, enc, lifter, feats, metas, loss_dirs = tf._setup()          # shipped structure, reduced grid (25 x 25 x 7)
        base = tf._train_pass(enc, lifter, feats, metas, loss_dirs)
        with torch.no_grad():
            base_inf = [o.clone() for o in enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']]
        enc.row_shard = True                                            # every plane's rows split over the two ranks
        got = tf._train_pass(enc, lifter, feats, metas, loss_dirs)
        with torch.no_grad():
            got_inf = [o.clone() for o in enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']]
        plan = enc._row_shard_plan
        if plan.world_size != ws or plan.n_local >= sum(plan.sizes) or min(plan.local_sizes) < 1:
            msgs.append(f"no sharding happened: {plan.local_sizes} of {plan.sizes}")
        for i, (a, b) in enumerate(zip(base[0], got[0])):
            if (a - b).abs().max().item() > 1e-5 * max(1.0, a.abs().max().item()):
                msgs.append(f"train forward plane {i}: max diff {(a - b).abs().max().item():.3e}")
        for i, (a, b) in enumerate(zip(base_inf, got_inf)):
            if (a - b).abs().max().item() > 1e-5 * max(1.0, a.abs().max().item()):
                msgs.append(f"inference plane {i}: max diff {(a - b).abs().max().item():.3e}")
        if abs(base[1].item() - got[1].item()) > 1e-6 * abs(base[1].item()) + 1e-9:
            msgs.append(f"loss {base[1].item()} vs {got[1].item()}")
        worst = 0.0
        for k, a in base[2].items():
            b = got[2][k]
            e = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30)
            worst = max(worst, e)
            if e > 1e-4:      # float32 sums in a different order (row blocks, all-reduced partial sums)
                msgs.append(f"grad {k}: {e:.3e} of its scale")
        # ---- the two switches together: row-sharded encoder -> ray-sharded head -> losses, one training step ----
        # (the encoder's LAST all-gather passes its gradient through un-reduced: it must arrive complete and identical on
        # every rank from the head's all-reduced volume gradient)
        from selfocc_amd.registry import MODELS, OPENOCC_LOSS
        import selfocc_amd.loss  # noqa: F401
        import json
        cfg = json.load(open(os.path.join(tf.G, "encoder_full_cfg.json")))
        l2i = np.asarray(metas[0]['lidar2img'], dtype=np.float64)
        metas2 = [dict(metas[0], img2lidar=np.linalg.inv(l2i), temImg2lidar=np.linalg.inv(l2i))]
        torch.manual_seed(3)
        head = MODELS.build(dict(type='NeuSHead', roi_aabb=[-40.0, -40.0, -1.0, 40.0, 40.0, 5.4], resolution=1.6, near_plane=0.0,
                                 far_plane=1e10, num_samples=64, num_samples_importance=0, num_up_sample_steps=0, base_variance=4,
                                 beta_init=0.1, beta_hand_tune=False, use_numerical_gradients=False, sample_gradient=True,
                                 return_uniform_sdf=False, return_second_grad=True, use_compact_2nd_grad=True, return_sem=True,
                                 ray_sample_mode='fixed', ray_number=[7, 10], ray_img_size=[96, 200], trans_kw='temImg2lidar',
                                 render_bkgd='white', mapping_args=cfg['encoder']['mapping_args'], embed_dims=96, color_dims=8,
                                 density_layers=2, sh_deg=0, sh_act='relu', two_split=False, tpv=True, single_jitter=True)).to(tf.D0).eval()
        with torch.no_grad():
            head.model.field.density_net[-1].bias[0] = 0.5
        loss_fn = OPENOCC_LOSS.build(dict(type='MultiLoss', sync_items=False, loss_cfgs=[
            dict(type='EikonalLoss', weight=0.1), dict(type='SecondGradLoss', weight=0.01)]))
        gcoef = torch.Generator().manual_seed(9)
        cd = torch.randn(1, 6, 70, generator=gcoef).to(tf.D0)
        cc = torch.randn(1, 6, 70, 3, generator=gcoef).to(tf.D0)

        def step(shard):
            enc.row_shard = head.ray_shard = shard
            os.environ['eval'] = 'false'
            for p in list(enc.parameters()) + list(lifter.parameters()) + list(head.parameters()):
                p.grad = None
            fs = [f.detach().clone().requires_grad_(True) for f in feats]
            rep = enc(lifter(fs)['representation'], ms_img_feats=fs, metas=metas2)['representation']
            out = head(rep, metas2, global_iter=0)
            total, _ = loss_fn(dict(out, metas=metas2))
            total = total + (out['ms_depths'][0] * cd).mean() + (out['ms_colors'][0] * cc).mean()
            total.backward()
            g = {('enc', n): p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
            g.update({('lift', n): p.grad.clone() for n, p in lifter.named_parameters() if p.grad is not None})
            g.update({('head', n): p.grad.clone() for n, p in head.named_parameters() if p.grad is not None})
            g.update({('feat', str(i)): f.grad.clone() for i, f in enumerate(fs)})
            return total.detach(), g
        l0, g0 = step(False)
        l1, g1 = step(True)
        if abs(l0.item() - l1.item()) > 1e-4 * abs(l0.item()) + 1e-7:
            msgs.append(f"combined: loss {l0.item()} vs {l1.item()}")
        if set(g0) != set(g1):
            msgs.append(f"combined: gradient sets differ {sorted(set(g0) ^ set(g1))[:5]}")
        worst2 = 0.0
        for k in g0:
            if k in g1:
                e = (g0[k] - g1[k]).abs().max().item() / max(g0[k].abs().max().item(), 1e-30)
                worst2 = max(worst2, e)
                if e > 2e-4:
                    msgs.append(f"combined grad {k}: {e:.3e} of its scale")
        ret[rank] = msgs
        ret[f'worst{rank}'] = (worst, worst2)
    except Exception as e:   # surface the failure in the parent
        import traceback
        ret[rank] = [f"exception: {e!r}\n{traceback.format_exc()}"]
    finally:
        dist.destroy_process_group()


def _bev_worker(rank, ws, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import copy, json
        from selfocc_amd.registry import MODELS
        import selfocc_amd.model  # noqa: F401
        G = os.path.join(os.path.dirname(__file__), "golden")
        D0 = torch.device("cuda:0")
        z = np.load(os.path.join(G, "bev_encoder.npz"))
        cfg = json.load(open(os.path.join(G, "bev_encoder_cfg.json")))
        enc = MODELS.build(dict(type='BEVFormerEncoder', **copy.deepcopy(cfg['encoder'])))
        lifter = MODELS.build(dict(type='BEVQueryLifter', **cfg['lifter']))
        enc.load_state_dict({k[4:]: torch.tensor(v) for k, v in z.items() if k.startswith('enc.')}, strict=True)
        lifter.load_state_dict({k[5:]: torch.tensor(v) for k, v in z.items() if k.startswith('lift.')}, strict=True)
        enc, lifter = enc.to(D0).eval(), lifter.to(D0).eval()
        metas = [dict(lidar2img=z['lidar2img'], img_shape=tuple(cfg['img_shape']))]
        coef = None
        msgs = []

        def step(shard):
            nonlocal coef
            enc.row_shard = shard
            for p in list(enc.parameters()) + list(lifter.parameters()):
                p.grad = None
            fs = [torch.tensor(z['feat0']).to(D0).requires_grad_(True), torch.tensor(z['feat1']).to(D0).requires_grad_(True)]
            with torch.no_grad():
                inf = enc(lifter(fs)['representation'], ms_img_feats=fs, metas=metas)['representation'].clone()
            out = enc(lifter(fs)['representation'], ms_img_feats=fs, metas=metas)['representation']
            if coef is None:
                coef = torch.randn(out.shape, generator=torch.Generator().manual_seed(4)).to(D0)
            (out * coef).mean().backward()
            g = {('enc', n): p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
            g.update({('lift', n): p.grad.clone() for n, p in lifter.named_parameters() if p.grad is not None})
            g.update({('feat', str(i)): f.grad.clone() for i, f in enumerate(fs)})
            return inf, out.detach(), g
        i0, o0, g0 = step(False)
        i1, o1, g1 = step(True)
        if enc._row_shard_plan.n_local >= sum(enc._row_shard_plan.sizes):
            msgs.append("no sharding happened")
        for nm, a, b in (("inference", i0, i1), ("train forward", o0, o1)):
            if (a - b).abs().max().item() > 1e-5 * max(1.0, a.abs().max().item()):
                msgs.append(f"{nm}: max diff {(a - b).abs().max().item():.3e}")
        if set(g0) != set(g1):
            msgs.append(f"gradient sets differ: {sorted(set(g0) ^ set(g1))[:5]}")
        worst = 0.0
        for k in g0:
            if k in g1:
                e = (g0[k] - g1[k]).abs().max().item() / max(g0[k].abs().max().item(), 1e-30)
                worst = max(worst, e)
                if e > 1e-4:
                    msgs.append(f"grad {k}: {e:.3e} of its scale")
        ret[rank] = msgs
        ret[f'worst{rank}'] = worst
    except Exception as e:   # surface the failure in the parent
        import traceback
        ret[rank] = [f"exception: {e!r}\n{traceback.format_exc()}"]
    finally:
        dist.destroy_process_group()


def test_row_sharded_bev_encoder_equals_unsharded_world2(hip):
    """BEVFormerEncoder(row_shard=True) on two ranks (the reference's own state dict, tests/golden/bev_encoder.npz): forward
    (inference and training route) and every gradient equal the unsharded encoder's."""
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_bev_worker, args=(r, ws, port, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for r in range(ws):
        assert ret.get(r) == [], f"rank {r}: {ret.get(r)}"
    print("worst relative gradient difference (BEV)", [ret.get(f'worst{r}') for r in range(ws)])


def test_row_sharded_encoder_equals_unsharded_world2(hip):
    """TPVFormerEncoder(row_shard=True) on two ranks (queries sharded, values replicated, one all-gather
    of the planes per layer): forward planes (inference and training route) and EVERY gradient — parameters, lifter queries,
    FPN maps — equal the unsharded encoder's on both ranks; then the two switches together (row-sharded encoder -> ray-sharded
    NeuSHead -> losses): loss and every gradient of one training step equal the unsharded step's."""
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_enc_worker, args=(r, ws, port, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for r in range(ws):
        assert ret.get(r) == [], f"rank {r}: {ret.get(r)}"
    print("worst relative gradient difference (encoder alone, encoder + ray-sharded head + losses)", [ret.get(f'worst{r}') for r in range(ws)])


@pytest.mark.parametrize("shard", ["frames", "rays"])
def test_bench_two_ranks_on_one_gpu(shard):
    """bench.py's N > 1 path exactly as the driver launches it (torch.distributed.run, 2 ranks), both ranks on cuda:0 over
    gloo (SELFOCC_BENCH_SHARE_GPU=1: this box has one GPU; the driver's runs use one GPU per rank over RCCL): barriers,
    async loss all-reduce, max-over-ranks timing, the one JSON line of rank 0."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SELFOCC_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sk:                                 # a port nobody listens on right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--preheat", "2", "--shard", shard, "--no-cpu-baseline", "--no-extras", "--no-hotpath"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["value"] > 0
    assert line["scaling"] == ("strong" if shard == "rays" else "weak")
    seen = line["ranks_seen"]
    assert seen["world_size"] == 2 and seen["backend"] == "gloo" and {r_["rank"] for r_ in seen["ranks"]} == {0, 1}
    assert 1.0 < line["allreduced_mean_depth_m"] < 60.0
    rays = 6 * 450 * 800
    per_step = rays if shard == "rays" else 2 * rays
    assert abs(line["value"] - per_step / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


def test_bench_plain_python_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun (round-4 review: it silently ran one rank and printed n_gpus = 1): bench.py
    re-executes itself under torch.distributed.run, and the one line it prints has n_gpus == --gpus."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["SELFOCC_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--preheat", "2",
           "--no-cpu-baseline", "--no-extras", "--no-hotpath"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"]["world_size"] == 2 and line["value"] > 0
    # ... and a WORLD_SIZE that contradicts --gpus is an error, not a line
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline",
                          "--no-extras", "--no-hotpath"], cwd=root, env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def _enc_shipped_worker(rank, ws, port, ret):
    """TPVFormerEncoder(row_shard=True) built from the SHIPPED nuscenes_occ config at its full size (257 x 257 x 25, 6 cameras,
    78 899 queries): the fast paths run with local-row query counts (camera-loop kernels, fused self-attention, banded
    scatter), first in eval mode against the unsharded encoder, then one TRAINING-mode step (dropout 0.1: the fused
    dropout + residual path under sharding) whose outputs and gradients must be finite and identical in shape."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "scripts"))
        import hotpath_common as hc
        d = torch.device("cuda:0")
        msgs = []
        torch.manual_seed(0)
        cfg = hc.shipped("nuscenes_occ")
        lifter, enc, _head, _ = hc.build(cfg, d)
        img = tuple(cfg['img_size'])
        c2w, l2i, K = hc.ring_cameras(6, img, 1266.0)
        metas = [dict(lidar2img=l2i, img2lidar=c2w, img_shape=img)]
        g = torch.Generator().manual_seed(5)
        feats = [torch.randn(1, 6, 96, -(-img[0] // s_), -(-img[1] // s_), generator=g).to(d) for s_ in (8, 16, 32, 64)]
        enc.eval()
        with torch.no_grad():
            base = [o.clone() for o in enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']]
            enc.row_shard = True
            got = enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']
        plan = enc._row_shard_plan
        if plan.world_size != ws or plan.n_local >= sum(plan.sizes):
            msgs.append(f"no sharding happened: {plan.local_sizes} of {plan.sizes}")
        chk = torch.tensor([float(b.double().abs().sum()) for b in base]
                           + [float(sum(p.detach().double().abs().sum() for p in enc.parameters()))], dtype=torch.float64)
        both = [torch.empty_like(chk) for _ in range(ws)]
        dist.all_gather(both, chk)
        if not all(torch.equal(both[0], o) for o in both):
            msgs.append(f"the ranks' UNSHARDED encoders differ (not the same weights / inputs?): {[o.tolist() for o in both]}")
        for i, (a, b) in enumerate(zip(base, got)):
            lo, hi = plan.local[i]
            mine = torch.zeros(a.shape[1], dtype=torch.bool, device=a.device)
            mine[lo:hi] = True
            sc = max(1.0, a.abs().max().item())
            e_own = (a - b)[:, mine].abs().max().item() / sc
            e_rem = (a - b)[:, ~mine].abs().max().item() / sc
            if max(e_own, e_rem) > 1e-5:
                msgs.append(f"shipped-size inference plane {i}: own rows {e_own:.3e}, rows gathered from the other rank {e_rem:.3e}")
        # one training-mode step under sharding (dropout on).  init_weights() zeroes the sampling_offsets / attention_weights
        # Linears, the only consumers of the positional encodings: their gradient would be exactly 0 — perturb every
        # parameter (same draw on both ranks) so that "zero gradient" means "not reached"
        gen = torch.Generator().manual_seed(17)
        with torch.no_grad():
            for p_ in enc.parameters():
                p_.add_((0.02 * torch.randn(p_.shape, generator=gen)).to(p_.device))
        enc.train()
        fs = [f.detach().clone().requires_grad_(True) for f in feats]
        out = enc(lifter(fs)['representation'], ms_img_feats=fs, metas=metas)['representation']
        sum((o * o).mean() for o in out).backward()
        torch.cuda.synchronize()
        if [tuple(o.shape) for o in out] != [tuple(b.shape) for b in base]:
            msgs.append("training-mode output shapes differ")
        bad = [n for n, p in list(enc.named_parameters()) + list(lifter.named_parameters())
               if p.grad is None or not torch.isfinite(p.grad).all() or p.grad.abs().max() == 0]
        if bad:
            msgs.append(f"missing / non-finite / zero gradients: {bad[:6]}")
        if not all(torch.isfinite(f.grad).all() and f.grad.abs().max() > 0 for f in fs):
            msgs.append("feature gradients missing")
        try:        # batch size 2 must raise a readable error, not an opaque reshape failure
            enc([torch.cat([q, q]) for q in lifter(feats)['representation']], ms_img_feats=[torch.cat([f, f]) for f in feats],
                metas=metas * 2)
            msgs.append("bs = 2 under row_shard did not raise")
        except NotImplementedError:
            pass
        ret[rank] = msgs
    except Exception as e:   # surface the failure in the parent
        import traceback
        ret[rank] = [f"exception: {e!r}\n{traceback.format_exc()}"]
    finally:
        dist.destroy_process_group()


def test_row_sharded_encoder_at_the_shipped_size_and_in_training_mode(hip):
    ws = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_enc_shipped_worker, args=(r, ws, port, ret)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for r in range(ws):
        assert ret.get(r) == [], f"rank {r}: {ret.get(r)}"
