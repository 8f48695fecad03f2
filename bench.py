#!/usr/bin/env python
"""bench.py — SelfOcc hot path on MI355X: rendered rays / second.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: one full
nuScenes-sized frame of BASELINE.json configs[1] — 6 cameras x 450x800 rays, 128
samples / ray, 200x200x16 volume — rendered by selfocc_render_fwd (ray generation,
AABB clip, sampling, trilinear SDF/colour/semantic lookup, NeuS alpha, compositing).
Inputs (volume, camera matrices) are resident in HBM before the timed region.
N > 1: one process per GPU (torchrun env), every rank renders its own frame (rays are
independent units: no data-path collective) and the ranks all-reduce the scalar
rendered-depth loss over RCCL each step, as north_star describes => weak scaling.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the render
kernel is the only kernel in a step); `cpu_baseline` times the torch-op port of the
reference's CPU render path (oracle/torch_port.py) on a bounded sample, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(vol, n_rays, n_sem):
    """every distinct input byte once + every API-visible output byte once.
    Rays are generated in-kernel from the 6 camera matrices (I = 0 B / ray)."""
    v = vol.sdf.numel() * 4 + (0 if vol.feat is None else vol.feat.numel() * vol.feat.element_size())
    per_ray_out = 4 * 5 + (12 if vol.n_rgb else 0) + 4 * n_sem  # depth acc max_depth nears fars rgb sem
    return v + n_rays * per_ray_out, v, per_ray_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", type=int, default=1, choices=[1, 4, 25],
                    help="volume channels: 1 = sdf only (config/nuscenes/nuscenes_depth.py, color_dims=0: the "
                         "eval_depth.py path the reference's README quotes) | 4 sdf+rgb | 25 sdf+rgb+21 sem (nuscenes_occ)")
    ap.add_argument("--feat-dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--exact", action="store_true", help="canonical IEEE path (bit-exact with the oracle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-hotpath", action="store_true",
                    help="skip the whole-path stage timings (scripts/bench_hotpath_*.py) reported under \"hot_path\"")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # SELFOCC_BENCH_SHARE_GPU=1 (testing only): every rank uses cuda:0 over gloo, so that the N > 1 code
    # path can be exercised on a 1-GPU box; the driver's real runs use one GPU per rank over RCCL
    share = os.environ.get("SELFOCC_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    from selfocc_amd import synthetic as sy
    from selfocc_amd.render import render_rays, RaySet

    name = "cfg2"
    n_rgb, n_sem = {1: (0, 0), 4: (3, 0), 25: (3, 21)}[args.channels]
    fdt = torch.float32 if args.feat_dtype == "f32" else torch.bfloat16
    vol_cpu = sy.make_volume(name, n_rgb=n_rgb, n_sem=n_sem, feat_dtype=fdt, seed=rank)
    rays_cpu = sy.make_rays(name, seed=rank)
    cfg = sy.make_render_config(name, inv_s=20.0, exact=args.exact)
    vol = vol_cpu.to(dev)
    rays = RaySet(img2lidar=rays_cpu.img2lidar.to(dev), nx=rays_cpu.nx, ny=rays_cpu.ny,
                  sx=rays_cpu.sx, sy=rays_cpu.sy)
    n_rays = rays.n_rays
    out = render_rays(vol, rays, cfg)  # allocates outputs once
    # ray-sharded ranks all-reduce the rendered-depth loss (north_star): one 4-byte RCCL all-reduce per
    # step, issued asynchronously into its own slot so that step i+1's render never waits for it
    losses = torch.zeros(args.steps + args.warmup, device=dev)
    pending = []

    def step(i):
        render_rays(vol, rays, cfg, outputs=out)
        if world > 1:
            slot = losses[i:i + 1]
            torch.mean(out['depth'], dim=0, keepdim=True, out=slot)
            pending.append(dist.all_reduce(slot, async_op=True))

    def fence():
        while pending:
            pending.pop().wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = elapsed / args.steps * 1e3
    value = n_rays * world * args.steps / elapsed

    # ---- roofline of the dominant kernel: HIP events on the launch stream, kernel only ----
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    for e0, e1 in evs:
        e0.record()
        render_rays(vol, rays, cfg, outputs=out)
        e1.record()
    torch.cuda.synchronize()
    kern_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
    alg_bytes, vol_bytes, per_ray_out = algorithmic_bytes(vol, n_rays, n_sem)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    touch_bytes = n_rays * cfg.n_samples * 8 * (4 + (0 if vol.feat is None else (n_rgb + n_sem) * vol.feat.element_size()))
    roofline = {
        "bound": "hbm", "kernel": f"render_fwd_pixgrid<NF={vol.feat.shape[3] if vol.feat is not None else 0}>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
        "traffic": None,  # rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE per launch: see profiles/ (filled by hand per round)
        "kernel_ms": round(kern_ms, 4), "algorithmic_bytes": alg_bytes,
        "note": ("compulsory bytes = volume once + per-ray outputs; the march is gather/VALU bound "
                 "(volume <= 64 MB sits in L2 / Infinity Cache), touched bytes through L1 per launch = %d" % touch_bytes),
        "touch_GBps": round(touch_bytes / (kern_ms * 1e-3) / 1e9, 1),
    }
    # HBM-side bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
    # scripts/pmc.sh); committed per round under profiles/ because PMC collection cannot run inside bench.py
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        key = f"c{args.channels}_{args.feat_dtype}_{'exact' if args.exact else 'fast'}"
        rec = json.load(open(pmc_file)).get(key)
        if rec:
            roofline["traffic"] = rec["traffic_bytes"]
            roofline["traffic_note"] = rec["note"]
            if "valu_wave_insts" in rec:   # secondary, compute-side view of the same kernel
                peak = 256 * 4 * 32 * 2.4e9 / 1e12          # fp32 vector lane-ops / s (157.3 TFLOP/s = 2 flop x this)
                ach = rec["valu_wave_insts"] * 64 / (kern_ms * 1e-3) / 1e12
                roofline["valu"] = {"achieved_Tlaneops": round(ach, 2), "peak_Tlaneops": round(peak, 2),
                                    "frac": round(ach / peak, 3),
                                    "note": "SQ_INSTS_VALU x 64 lanes / kernel time vs 256 CU x 4 SIMD x 32 lanes x 2.4 GHz"}

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        def time_variant(c, dt, exact, k=10):
            nr, ns = {1: (0, 0), 4: (3, 0), 25: (3, 21)}[c]
            v = sy.make_volume(name, n_rgb=nr, n_sem=ns, feat_dtype=dt, seed=0).to(dev)
            cf = sy.make_render_config(name, inv_s=20.0, exact=exact)
            o = render_rays(v, rays, cf)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(k):
                render_rays(v, rays, cf, outputs=o)
            b.record()
            torch.cuda.synchronize()
            return round(n_rays / (a.elapsed_time(b) / k * 1e-3), 1)
        extras = {
            "rays_per_s_c1_f32": time_variant(1, torch.float32, False),
            "rays_per_s_c1_f32_exact": time_variant(1, torch.float32, True, k=3),
            "rays_per_s_c4_f32": time_variant(4, torch.float32, False),
            "rays_per_s_c25_f32": time_variant(25, torch.float32, False),
            "rays_per_s_c25_bf16": time_variant(25, torch.bfloat16, False),
            "rays_per_s_c25_f32_exact": time_variant(25, torch.float32, True, k=3),
        }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the reference's CPU render path (torch F.grid_sample + NeuS compositing) on the same frame, in
        # README-sized chunks of 90 000 rays (neus_head.py:329-385) until ~12 s of CPU work are spent;
        # oracle/ is used here ONLY as the timed baseline
        from oracle import torch_port as tp
        ex = sy.explicit_rays(rays_cpu)
        dc = vol_cpu.to_reference_layout()
        cores = torch.get_num_threads()
        tp.render_port(vol_cpu.mapping, dc, n_rgb, n_sem, ex.origins[:2000], ex.dirs[:2000], ex.dir_norm[:2000], cfg)
        chunk, done, spent = 90_000, 0, 0.0
        while done < n_rays and spent < 12.0:
            sl = slice(done, min(n_rays, done + chunk))
            c0 = time.perf_counter()
            tp.render_port(vol_cpu.mapping, dc, n_rgb, n_sem, ex.origins[sl], ex.dirs[sl], ex.dir_norm[sl], cfg, chunk=chunk)
            spent += time.perf_counter() - c0
            done = sl.stop
        cpu_baseline = {"value": round(done / spent, 1), "unit": "rays/s", "cores": cores, "kind": "port",
                        "sample": f"first {done} rays of the same cfg2 frame (chunks of 90000), C={args.channels}, "
                                  f"torch CPU F.grid_sample + NeuS compositing, {spent:.1f} s"}

    hot_path = None
    if rank == 0 and world == 1 and not args.no_hotpath and not args.no_extras:
        # the rest of the hot path at the reference's shipped shapes (no image backbone), outside the timed
        # region and in their own processes: lifter + encoder + field volume + render of one nuscenes_depth
        # evaluation frame, and one nuscenes_occ training iteration (forward, five losses, backward)
        import subprocess
        torch.cuda.empty_cache()
        hot_path = {}
        here = os.path.dirname(os.path.abspath(__file__))
        for key, script in (("eval_frame_nuscenes_depth_ms", "bench_hotpath_eval.py"),
                            ("train_iteration_nuscenes_occ_ms", "bench_hotpath_train.py")):
            try:
                r = subprocess.run([sys.executable, os.path.join(here, "scripts", script)], capture_output=True,
                                   text=True, timeout=240)
                last = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
                hot_path[key] = json.loads(last[-1]) if last else {"error": (r.stderr or "no output")[-300:]}
            except Exception as e:   # never let the side measurement break the bench line
                hot_path[key] = {"error": repr(e)[:300]}

    if rank == 0:
        line = {
            "metric": "rendered rays/sec (6-cam 450x800, 128 samples/ray)",
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 6 cams x 450x800 rays, 128 samples/ray, volume 200x200x16",
                       "volume_channels": args.channels, "feat_storage": args.feat_dtype,
                       "rays_per_step_per_gpu": n_rays, "inv_s": 20.0,
                       "path": "exact" if args.exact else "fast", "sharding": f"frame-per-rank x{world}"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if extras:
            line["extras"] = extras
        if hot_path:
            line["hot_path"] = hot_path
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
