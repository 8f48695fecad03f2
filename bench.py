#!/usr/bin/env python
"""bench.py — SelfOcc hot path on MI355X: rendered rays / second.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: one full
nuScenes-sized frame of BASELINE.json configs[1] — 6 cameras x 450x800 rays, 128
samples / ray, 200x200x16 volume — rendered by selfocc_render_fwd (ray generation,
AABB clip, sampling, trilinear SDF/colour/semantic lookup, NeuS alpha, compositing).
Inputs (volume, camera matrices) are resident in HBM before the timed region.
N > 1: one process per GPU — under torch.distributed.run as the driver launches it, or started by bench.py itself when it
is run as plain `python bench.py --gpus N` (it re-executes under torch.distributed.run).  Default (`--shard frames`): the global ray batch is N
frames, every rank marches one frame's worth of rays (rays are independent units: no data-path
collective) and the ranks all-reduce the scalar rendered-depth loss over RCCL each step, as
north_star describes => weak scaling.  `--shard rays`: ONE frame is split into
row blocks over the ranks (selfocc_amd.dist.shard_rays, the mode NeuSHead(ray_shard=True) runs) =>
strong scaling; at N > 1 the default run also reports it under "strong_scaling".

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
L1_PEAK_GBPS = 256 * 64 * 2.4   # vector L1 (TCP): 64 B / clk / CU x 256 CUs x 2.4 GHz = 39.3 TB/s


def algorithmic_bytes(vol, n_rays, n_sem):
    """every distinct input byte once + every API-visible output byte once.
    Rays are generated in-kernel from the 6 camera matrices (I = 0 B / ray)."""
    v = vol.sdf.numel() * 4 + (0 if vol.feat is None else vol.feat.numel() * vol.feat.element_size())
    per_ray_out = 4 * 5 + (12 if vol.n_rgb else 0) + 4 * n_sem  # depth acc max_depth nears fars rgb sem
    return v + n_rays * per_ray_out, v, per_ray_out


def kernel_source_hash():
    """sha1 (12 hex) of what the render kernel is compiled from — render_fwd.hip, so_device.h and the part of the
    public header it sees (the SO_FLAG_* enum and struct so_render_args): ties a PMC record in
    profiles/pmc_traffic.json to the kernel it was measured on (scripts/pmc.sh writes the same hash next to the
    counters).  Header changes for other entry points (MSDA, LayerNorm ...) do not change the render kernel."""
    import hashlib
    import re
    h = hashlib.sha1()
    for f in ("selfocc_amd/csrc/render_fwd.hip", "selfocc_amd/csrc/so_device.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    hdr = open(os.path.join(ROOT, "include", "selfocc_hip.h")).read()
    m = re.search(r"typedef struct so_render_args \{.*?\} so_render_args;", hdr, re.S)
    h.update(m.group(0).encode())
    h.update("\n".join(l for l in hdr.splitlines() if "SO_FLAG_" in l).encode())
    return h.hexdigest()[:12]


BWD_SOURCES = ("render_bwd.hip", "render_train.hip", "field_bwd_b3.hip", "field.hip", "msda.hip", "msda_device.h", "so_device.h")
MSDA_SOURCES = ("msda.hip", "msda_device.h", "so_device.h")


def sources_hash(files):
    """sha1 (12 hex) of csrc files: keys profiles/pmc_bwd.json / pmc_msda.json to the kernels they were measured on"""
    import hashlib
    h = hashlib.sha1()
    for f in files:
        h.update(open(os.path.join(ROOT, "selfocc_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preheat", type=int, default=150,
                    help="untimed launches before the warm-up steps (clock ramp; 0 = off), reported as config.preheat_steps")
    ap.add_argument("--channels", type=int, default=1, choices=[1, 4, 25],
                    help="volume channels: 1 = sdf only (config/nuscenes/nuscenes_depth.py, color_dims=0: the "
                         "eval_depth.py path the reference's README quotes) | 4 sdf+rgb | 25 sdf+rgb+21 sem (nuscenes_occ)")
    ap.add_argument("--feat-dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--exact", action="store_true", help="canonical IEEE path (bit-exact with the oracle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--shard", default="frames", choices=["frames", "rays"],
                    help="N > 1: frames = one frame per rank (weak scaling, default) | rays = one frame split into row "
                         "blocks over the ranks (strong scaling, SURVEY cfg3)")
    ap.add_argument("--inv-s", type=float, default=20.0)
    ap.add_argument("--no-skip", action="store_true", help="fast path without free-space skipping (A/B)")
    ap.add_argument("--no-face-safe", action="store_true", help="fast path without canonical cell selection near voxel faces (A/B)")
    ap.add_argument("--no-hotpath", action="store_true",
                    help="skip the whole-path stage timings (scripts/bench_hotpath_*.py) reported under \"hot_path\"")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (the reference spawns its own, train.py:400-401) by
        # re-executing this command under torch.distributed.run — one process per GPU, rendezvous on 127.0.0.1.  A line whose
        # n_gpus differs from --gpus is never printed.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: the line's n_gpus must be the number asked for"
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
    from selfocc_amd.dist import shard_rays

    name = "cfg2"
    n_rgb, n_sem = {1: (0, 0), 4: (3, 0), 25: (3, 21)}[args.channels]
    fdt = torch.float32 if args.feat_dtype == "f32" else torch.bfloat16
    split = args.shard == "rays" and world > 1
    seed = 0 if split else rank                 # ray-sharded ranks work on the SAME frame
    vol_cpu = sy.make_volume(name, n_rgb=n_rgb, n_sem=n_sem, feat_dtype=fdt, seed=seed)
    rays_cpu = sy.make_rays(name, seed=seed)
    cfg = sy.make_render_config(name, inv_s=args.inv_s, exact=args.exact, skip=not args.no_skip,
                                face_safe=not args.no_face_safe)
    vol = vol_cpu.to(dev)
    full_rays = RaySet(img2lidar=rays_cpu.img2lidar.to(dev), nx=rays_cpu.nx, ny=rays_cpu.ny,
                       sx=rays_cpu.sx, sy=rays_cpu.sy)
    rays = shard_rays(full_rays, rank, world) if split else full_rays
    n_rays = rays.n_rays
    rays_per_step_all_ranks = full_rays.n_rays if split else n_rays * world
    out = render_rays(vol, rays, cfg)  # allocates outputs once
    # ray-sharded ranks all-reduce the rendered-depth loss (north_star): one 4-byte RCCL all-reduce per
    # step, issued asynchronously into its own slot so that step i+1's render never waits for it
    losses = torch.zeros(args.steps + args.warmup, device=dev)
    pending = []

    def step(i):
        render_rays(vol, rays, cfg, outputs=out)
        if world > 1:
            # the local SUM of the rendered depths (shards may differ by a row: sums, not means, add up); the division by
            # the global ray count is applied once, when the line is written (one kernel per step, not two)
            slot = losses[i:i + 1]
            torch.sum(out['depth'], dim=0, keepdim=True, out=slot)
            pending.append(dist.all_reduce(slot, async_op=True))

    def fence():
        while pending:
            pending.pop().wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed pre-heat BEFORE the W warm-up steps: a step is 0.4 ms, so W = 5 leaves the chip on its idle clocks when the
    # timed region starts (measured 5.3 - 5.8 G rays/s from box to box); ~50 ms of the same launch (no collectives) puts
    # the timed region at the steady state a 90-minute eval_depth run sees.  Disclosed in config.preheat_steps.
    for _ in range(args.preheat):
        render_rays(vol, rays, cfg, outputs=out)
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
    value = rays_per_step_all_ranks * args.steps / elapsed

    # ---- roofline of the dominant kernel: HIP events on the launch stream, kernel only ----
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    for e0, e1 in evs:
        e0.record()
        render_rays(vol, rays, cfg, outputs=out)
        e1.record()
    torch.cuda.synchronize()
    per_launch = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    kern_ms = sum(per_launch) / len(per_launch)
    alg_bytes, vol_bytes, per_ray_out = algorithmic_bytes(vol, n_rays, n_sem)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    touch_bytes = n_rays * cfg.n_samples * 8 * (4 + (0 if vol.feat is None else (n_rgb + n_sem) * vol.feat.element_size()))
    roofline = {
        "bound": "hbm", "kernel": f"render_fwd_pixgrid<NF={vol.feat.shape[3] if vol.feat is not None else 0}>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
        "traffic": None,  # rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE per launch: see profiles/ (filled by hand per round)
        "kernel_ms": round(kern_ms, 4),
        "kernel_ms_min_median_max": [round(per_launch[0], 4), round(per_launch[len(per_launch) // 2], 4), round(per_launch[-1], 4)],
        "algorithmic_bytes": alg_bytes,
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
        if rec and rec.get("kernel_source_sha1") != kernel_source_hash():
            # counters taken on an older build of the kernel say nothing about this one
            roofline["traffic_note"] = (f"profiles/pmc_traffic.json[{key}] was measured on kernel source "
                                        f"{rec.get('kernel_source_sha1')} != current {kernel_source_hash()}: not reported")
            rec = None
        if rec:
            roofline["traffic"] = rec["traffic_bytes"]
            roofline["traffic_note"] = rec["note"]
            if "valu_wave_insts" in rec:   # secondary, compute-side view of the same kernel
                peak = 256 * 4 * 32 * 2.4e9 / 1e12          # fp32 vector lane-ops / s (157.3 TFLOP/s = 2 flop x this)
                ach = rec["valu_wave_insts"] * 64 / (kern_ms * 1e-3) / 1e12
                roofline["valu"] = {"achieved_Tlaneops": round(ach, 2), "peak_Tlaneops": round(peak, 2),
                                    "frac": round(ach / peak, 3),
                                    "note": "SQ_INSTS_VALU x 64 lanes / kernel time vs 256 CU x 4 SIMD x 32 lanes x 2.4 GHz"}
                roofline["valu_frac"] = round(ach / peak, 3)
            # The OPERATIVE bound of the march, from per-unit utilisations of the same PMC session — not from GRBM_TA_BUSY, which
            # counts the cycles in which ANY of the 256 texture addressers is busy (0.95 on this kernel; kept as
            # `ta_any_busy_frac` for continuity) and over-states the load (round-4 review):
            #   ta_util        = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8 XCDs)     the AVERAGE addresser's busy share of the kernel
            #   ta_unit_util   = TA_TA_BUSY_sum / 256 TAs / the same cycles
            #   l1_gather_frac = bytes gathered through the vector L1 / (64 B/clk/CU x 256 CUs x 2.4 GHz = 39.3 TB/s)
            # `frac` above stays the HBM number the contract asks for (the volume sits in L2 / MALL).
            if rec.get("ta_busy_cycles") and rec.get("gui_active_cycles"):
                roofline["ta_any_busy_frac"] = round(rec["ta_busy_cycles"] / rec["gui_active_cycles"], 3)
            if rec.get("ta_busy_avr") and rec.get("gui_active_cycles_ta_pass"):
                per_xcd = rec["gui_active_cycles_ta_pass"] / 8.0
                roofline["ta_util"] = round(rec["ta_busy_avr"] / per_xcd, 3)
                if rec.get("ta_ta_busy_sum"):
                    roofline["ta_unit_util"] = round(rec["ta_ta_busy_sum"] / 256.0 / per_xcd, 3)
            if rec.get("tcp_total_cache_accesses"):
                roofline["l1_accesses_per_launch"] = rec["tcp_total_cache_accesses"]
    roofline["l1_gather_frac"] = round(roofline["touch_GBps"] / L1_PEAK_GBPS, 3)
    utils = {"TA": roofline.get("ta_util"), "L1": roofline["l1_gather_frac"], "VALU": roofline.get("valu_frac")}
    if utils["TA"] is not None:
        top = max((v, k) for k, v in utils.items() if v is not None)
        # a pipe is "the bound" only when it is (nearly) saturated; otherwise the kernel is limited by instruction issue and
        # gather latency at the occupancy its registers allow, and the largest utilisation says how far from that pipe's roof
        roofline["bound_operative"] = top[1] if top[0] >= 0.8 else "issue/latency"
        roofline["bound_operative_util"] = {k: v for k, v in utils.items() if v is not None}

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        def time_variant(c, dt, exact, k=10, **kw):
            nr, ns = {1: (0, 0), 4: (3, 0), 25: (3, 21)}[c]
            v = sy.make_volume(name, n_rgb=nr, n_sem=ns, feat_dtype=dt, seed=0).to(dev)
            cf = sy.make_render_config(name, **{**dict(inv_s=args.inv_s, exact=exact), **kw})
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
            "rays_per_s_c1_f32_no_skip": time_variant(1, torch.float32, False, skip=False),
            "rays_per_s_c1_f32_no_face_safe": time_variant(1, torch.float32, False, face_safe=False),
            "rays_per_s_c1_f32_inv_s_200": time_variant(1, torch.float32, False, inv_s=200.0),
            "rays_per_s_c1_f32_exact": time_variant(1, torch.float32, True, k=3),
            "rays_per_s_c4_f32": time_variant(4, torch.float32, False),
            "rays_per_s_c25_f32": time_variant(25, torch.float32, False),
            "rays_per_s_c25_bf16": time_variant(25, torch.bfloat16, False),
            "rays_per_s_c25_f32_exact": time_variant(25, torch.float32, True, k=3),
        }

    # ---- parity of the timed configuration: the C oracle (float32 canonical order, OpenMP) on EVERY ray of the
    # frame, against the outputs of the very launch that was timed (oracle/ is used as the checker only) ----
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        c0 = time.perf_counter()
        ref = oracle.render_fwd(vol_cpu, rays_cpu, cfg)
        oracle_s = time.perf_counter() - c0
        got = {k: out[k].cpu() for k in ref if k in out}
        ok = ref['acc'] > 0.05
        rel = (got['depth'] - ref['depth']).abs() / ref['depth'].abs().clamp_min(1e-6)
        parity = {"checker": "oracle/oracle_render.c (float32 canonical order) on every ray of the timed frame",
                  "n_rays": int(rays_cpu.n_rays), "frac_rays_acc_gt_0.05": round(ok.float().mean().item(), 4),
                  "parity_frac_1e-4": round((rel[ok] < 1e-4).float().mean().item(), 6),
                  "depth_max_rel_acc_gt_0.05": float(f"{rel[ok].max().item():.3e}"),
                  "depth_max_abs_all_rays_m": float(f"{(got['depth'] - ref['depth']).abs().max().item():.3e}"),
                  "acc_max_abs_all_rays": float(f"{(got['acc'] - ref['acc']).abs().max().item():.3e}"),
                  "excluded_rays": 0, "oracle_seconds": round(oracle_s, 2)}
        low = ~ok
        parity["low_acc_weighted_depth_err_over_far"] = float(
            f"{((got['depth'] - ref['depth']).abs() * ref['acc'] / ref['fars'].clamp_min(1e-6))[low].max().item():.3e}") if low.any() else 0.0
        parity["rule"] = ("acc > 0.05: |d - d_ref| < 1e-4 d_ref on every ray (parity_frac_1e-4 == 1); acc <= 0.05: "
                          "|d - d_ref| acc_ref <= 1e-4 far; every ray: |acc - acc_ref| <= 1e-4 "
                          "(tests/test_render_gpu.py::parity_report(strict=True) asserts the same rule)")
        parity["rule_holds"] = bool(parity["parity_frac_1e-4"] == 1.0 and parity["depth_max_rel_acc_gt_0.05"] < 1e-4
                                    and parity["low_acc_weighted_depth_err_over_far"] <= 1e-4
                                    and parity["acc_max_abs_all_rays"] <= 1e-4)
        for k in ('rgb', 'sem'):
            if k in ref:
                parity[k + "_max_abs_all_rays"] = float(f"{(got[k] - ref[k]).abs().max().item():.3e}")

    # ---- "the reference's batched-ray render path" on this GPU: the torch-op port of the reference's render
    # (F.grid_sample + autograd gradient + NeuS compositing, oracle/torch_port.py) executed with stock
    # PyTorch-ROCm ops on the MI355X in README-sized 90 000-ray chunks (neus_head.py:329-385).  Baseline only. ----
    gpu_torch_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_port as tp
        ex = sy.explicit_rays(rays_cpu)
        n_base = min(n_rays, 8 * 90_000)
        eo, ed, en = ex.origins[:n_base].to(dev), ex.dirs[:n_base].to(dev), ex.dir_norm[:n_base].to(dev)
        dc = vol.to_reference_layout()
        tp.render_port(vol.mapping, dc, n_rgb, n_sem, eo[:90_000], ed[:90_000], en[:90_000], cfg)   # warm-up
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        tp.render_port(vol.mapping, dc, n_rgb, n_sem, eo, ed, en, cfg, chunk=90_000)
        torch.cuda.synchronize()
        spent = time.perf_counter() - c0
        gpu_torch_baseline = {"value": round(n_base / spent, 1), "unit": "rays/s",
                              "kind": "torch-op port of the reference render on the same MI355X (stock PyTorch-ROCm "
                                      "ops, 90 000-ray chunks)",
                              "sample": f"first {n_base} rays of the same frame, C={args.channels}, {spent:.2f} s",
                              "speedup_of_value": round(value / (n_base / spent), 1)}
        del eo, ed, en, dc
        torch.cuda.empty_cache()

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the reference's CPU render path (torch F.grid_sample + NeuS compositing) on the same frame, in
        # README-sized chunks of 90 000 rays (neus_head.py:329-385) until ~12 s of CPU work are spent;
        # oracle/ is used here ONLY as the timed baseline
        from oracle import torch_port as tp
        ex = sy.explicit_rays(rays_cpu)
        dc = vol_cpu.to_reference_layout()
        # thread count: the best of a sweep on the GPU box's host (scripts/micro/cpu_port_threads.py, 256 hardware
        # threads): 8 / 16 / 32 / 64 / 128 / 256 torch threads = 53 / 55 / 53 / 41 / 17 / 5 k rays/s — torch's default (128)
        # oversubscribes these memory-bound ops
        default_threads = torch.get_num_threads()
        cores = min(16, default_threads)
        torch.set_num_threads(cores)
        tp.render_port(vol_cpu.mapping, dc, n_rgb, n_sem, ex.origins[:2000], ex.dirs[:2000], ex.dir_norm[:2000], cfg)
        chunk, done, spent = 90_000, 0, 0.0
        while done < n_rays and spent < 12.0:
            sl = slice(done, min(n_rays, done + chunk))
            c0 = time.perf_counter()
            tp.render_port(vol_cpu.mapping, dc, n_rgb, n_sem, ex.origins[sl], ex.dirs[sl], ex.dir_norm[sl], cfg, chunk=chunk)
            spent += time.perf_counter() - c0
            done = sl.stop
        torch.set_num_threads(default_threads)
        cpu_baseline = {"value": round(done / spent, 1), "unit": "rays/s", "cores": os.cpu_count(), "threads_used": cores, "kind": "port",
                        "sample": f"first {done} rays of the same cfg2 frame (chunks of 90000), C={args.channels}, "
                                  f"torch CPU F.grid_sample + NeuS compositing, {spent:.1f} s; {cores} torch threads = "
                                  f"the fastest of a 8..256 sweep on this host ({os.cpu_count()} hardware threads)"}
        # the plain-C restatement (OpenMP, all hardware threads) on the whole frame, for scale
        if parity:
            cpu_baseline["c_oracle_rays_per_s"] = round(rays_cpu.n_rays / parity["oracle_seconds"], 1)
            cpu_baseline["c_oracle_threads"] = os.cpu_count()

    hot_path = hot_path_detail = None
    if rank == 0 and world == 1 and not args.no_hotpath and not args.no_extras:
        # the rest of the hot path: ALL SEVEN shipped experiment configs (scripts/shipped_cfg/*.json = config/**/*.py dumped by
        # scripts/dump_shipped_configs.py), each built through the registries at its shipped shapes — one training iteration
        # (forward, the config's own loss list, backward) and the evaluation entry the reference's docs pair with it, with the
        # reference's eval-time overrides (scripts/hotpath_common.py: SHIPPED) — outside the timed region, in ONE process of its
        # own (scripts/bench_hotpath_all.py).  The line carries the totals; the per-stage split goes to the detail file.
        import subprocess
        torch.cuda.empty_cache()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_hotpath_all.py")], capture_output=True,
                               text=True, timeout=420)
            last = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
            hot_path_detail = json.loads(last[-1]) if last else {"error": (r.stderr or "no output")[-300:]}
        except Exception as e:   # never let the side measurement break the bench line
            hot_path_detail = {"error": repr(e)[:300]}
        if "error" in hot_path_detail:
            hot_path = hot_path_detail
        else:
            hot_path = {"unit": "ms (median), one training iteration (fwd + losses + bwd) / one frame of the config's eval entry, no backbone",
                        "train": {k: v["train"]["total_ms"] for k, v in hot_path_detail.items() if isinstance(v, dict) and "train" in v},
                        "eval": {k: v["eval"]["total_ms"] for k, v in hot_path_detail.items() if isinstance(v, dict) and "eval" in v},
                        "eval_entry": {k: v["eval"]["entry"] for k, v in hot_path_detail.items() if isinstance(v, dict) and "eval" in v}}

    roofline_msda = None
    if rank == 0 and world == 1 and not args.no_extras:
        # the lifter's kernels, where an HBM roofline is meaningful: algorithmic bytes / HIP-event time
        import subprocess
        torch.cuda.empty_cache()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_msda.py"), "--json"],
                               capture_output=True, text=True, timeout=240)
            last = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
            roofline_msda = json.loads(last[-1]) if last else {"error": (r.stderr or "no output")[-300:]}
        except Exception as e:
            roofline_msda = {"error": repr(e)[:300]}

    roofline_linear = None
    if rank == 0 and world == 1 and not args.no_extras:
        # the encoder's projections (selfocc_linear_fwd): float32 MFMA rate against its peak, next to hipBLASLt
        import subprocess
        torch.cuda.empty_cache()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_linear.py"), "--json"],
                               capture_output=True, text=True, timeout=240)
            last = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
            roofline_linear = json.loads(last[-1]) if last else {"error": (r.stderr or "no output")[-300:]}
        except Exception as e:
            roofline_linear = {"error": repr(e)[:300]}

    # ---- the training backward's scatter kernels: measured HBM-side bytes against their algorithmic bytes ----
    # (rocprofv3 --pmc cannot run inside bench.py: scripts/pmc_train_bwd.sh on the training iteration, committed per
    # round as profiles/pmc_bwd.json next to the text summary it was made from)
    roofline_bwd = None
    bwd_file = os.path.join(ROOT, "profiles", "pmc_bwd.json")
    if rank == 0 and world == 1 and os.path.exists(bwd_file):
        rec = json.load(open(bwd_file))
        # algorithmic bytes per launch at the nuscenes_occ training shape: the gradient the kernel produces, written once
        alg = {"render_bwd": 257 * 257 * 25 * 25 * 4,                       # d L / d (sdf + 24-channel feature) volume
               "field_volume_bwd": (257 * 257 + 2 * 25 * 257) * 96 * 4,     # the three plane gradients
               "msda_bwd_band_list": 6 * 25500 * 96 * 4}                    # d L / d value of one cross-attention call
        groups = {"render_bwd": ("render_bwd_kernel", "rb_brick_kernel", "rb_count_kernel"),
                  "field_volume_bwd": ("field_volume_bwd",), "msda_bwd_band_list": ("msda_bwd_band_list_kernel",)}
        roofline_bwd = {"source": "profiles/pmc_bwd.json (scripts/pmc_train_bwd.sh, training iteration at nuscenes_occ shapes)",
                        "measured_in_this_run": False,      # RECORDED counters + durations of that PMC session, not of this process
                        "recorded_round": rec.get("_round"),
                        # the record is keyed by a hash of the sources of the kernels it measured (like pmc_traffic.json):
                        "sources_sha1": rec.get("_sources_sha1"), "sources_match": rec.get("_sources_sha1") == sources_hash(BWD_SOURCES)}
        for name, pats in groups.items():
            ks = {k: v for k, v in rec.items()
                  if isinstance(v, dict) and any(k.startswith(p) for p in pats) and v.get("write_kb") is not None}
            if not ks:
                continue
            per_it = max(v["calls"] for v in ks.values()) or 1
            w = sum(v["write_kb"] * v["calls"] for v in ks.values()) / per_it * 1024
            f = sum((v["fetch_kb"] or 0) * v["calls"] for v in ks.values()) / per_it * 1024
            us = sum(v["avg_us"] * v["calls"] for v in ks.values()) / per_it
            roofline_bwd[name] = {"kernels": sorted(ks), "alg_MB": round(alg[name] / 1e6, 1), "write_MB": round(w / 1e6, 1),
                                  "fetch_MB": round(f / 1e6, 1), "write_over_alg": round(w / alg[name], 2),
                                  "us_per_launch": round(us, 1)}
            if len(ks) > 1:      # which kernel writes what (render_bwd: the brick kernel's writes ARE the gradient, the ray
                # kernel's are the per-sample records it hands over)
                roofline_bwd[name]["write_MB_by_kernel"] = {k.split("<")[0]: round(v["write_kb"] * 1024 / 1e6, 1) for k, v in ks.items()}

    strong = None
    if world > 1 and not split:
        # the same ranks, ONE frame split into row blocks (SURVEY cfg3), after the timed region
        f0 = sy.make_rays(name, seed=0)
        v0 = sy.make_volume(name, n_rgb=n_rgb, n_sem=n_sem, feat_dtype=fdt, seed=0).to(dev)
        fr = RaySet(img2lidar=f0.img2lidar.to(dev), nx=f0.nx, ny=f0.ny, sx=f0.sx, sy=f0.sy)
        mine = shard_rays(fr, rank, world)
        o2 = render_rays(v0, mine, cfg)
        dist.barrier(); torch.cuda.synchronize()
        c0 = time.perf_counter()
        acc2 = torch.zeros(args.steps, device=dev)
        hs = []
        for i in range(args.steps):
            render_rays(v0, mine, cfg, outputs=o2)
            torch.sum(o2['depth'], dim=0, keepdim=True, out=acc2[i:i + 1])
            hs.append(dist.all_reduce(acc2[i:i + 1], async_op=True))
        for h in hs:
            h.wait()
        dist.barrier(); torch.cuda.synchronize()
        t2 = torch.tensor([time.perf_counter() - c0], device=dev, dtype=torch.float64)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        strong = {"expected_bound": ("launch + all-reduce latency: a whole frame is ~0.37 ms on one GPU, so a 1/N row block "
                                     "(~0.37/N ms of kernel) sits next to ~10 us of launch and ~20-40 us of a small RCCL "
                                     "all-reduce per step; do not expect linear strong scaling of a single frame"),
                  "mode": "one frame split into row blocks over the ranks + all-reduce of the rendered-depth sum",
                  "rays_per_s": round(fr.n_rays * args.steps / t2.item(), 1),
                  "ms_per_frame": round(t2.item() / args.steps * 1e3, 4), "rays_per_rank": mine.n_rays}

    # which devices / which backend actually ran (the driver can check N distinct GPUs under RCCL)
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device_index": dev_index, "name": props.name,
          "uuid": str(getattr(props, "uuid", "")) or None, "pci_bus_id": getattr(props, "pci_bus_id", None)}
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, me)
        ranks_seen = {"backend": dist.get_backend(), "world_size": world, "ranks": seen,
                      "distinct_devices": len({(r["uuid"], r["pci_bus_id"], r["device_index"]) for r in seen})}
        if not share:   # a real multi-GPU line: one GPU per rank over RCCL, or no line at all
            assert ranks_seen["backend"] == "nccl", ranks_seen
            assert ranks_seen["distinct_devices"] == world, ranks_seen
    else:
        ranks_seen = {"backend": None, "world_size": 1, "ranks": [me], "distinct_devices": 1,
                      "launched_by": "torchrun" if "RANK" in os.environ else "python"}

    if rank == 0:
        # ---- the printed line stays under ~6 KB (the driver keeps `config`, `roofline`, `cpu_baseline` whole and the END of the
        # line): per-row / per-shape / per-stage tables go to gpurun_out/bench_detail.json; the parity of the timed frame and the
        # seven shipped configs' totals are in `config` (parsed) AND close the line (kept tail) ----
        detail = {"roofline_msda": roofline_msda, "roofline_linear": roofline_linear, "roofline_bwd": roofline_bwd,
                  "hot_path": hot_path_detail, "parity": parity, "extras": extras, "gpu_torch_baseline": gpu_torch_baseline,
                  "cpu_baseline": cpu_baseline, "roofline": dict(roofline)}
        cfg_out = {"workload": "BASELINE configs[1]: 6 cams x 450x800 rays, 128 samples/ray, volume 200x200x16",
                   "volume_channels": args.channels, "feat_storage": args.feat_dtype,
                   "rays_per_step_per_gpu": n_rays, "inv_s": args.inv_s, "preheat_steps": args.preheat,
                   "path": "exact" if args.exact else ("fast" + ("" if cfg.skip else ", no skip") + ("" if cfg.face_safe else ", no face_safe")),
                   "sharding": (f"one frame split by rows x{world}" if split else f"frame-per-rank x{world}")}
        par_c = None
        if parity:
            par_c = {"checker": "C oracle, every ray of the timed frame", "n_rays": parity["n_rays"], "rule_holds": parity["rule_holds"],
                     "depth_max_rel_acc_gt_0.05": parity["depth_max_rel_acc_gt_0.05"], "acc_max_abs": parity["acc_max_abs_all_rays"],
                     "depth_max_abs_m": parity["depth_max_abs_all_rays_m"],
                     "low_acc_weighted_depth_err_over_far": parity["low_acc_weighted_depth_err_over_far"],
                     "frac_rays_acc_gt_0.05": parity["frac_rays_acc_gt_0.05"]}
            cfg_out["parity_of_timed_frame"] = {k: par_c[k] for k in ("rule_holds", "depth_max_rel_acc_gt_0.05", "acc_max_abs", "depth_max_abs_m",
                                                                     "low_acc_weighted_depth_err_over_far")}
        if hot_path and "error" not in hot_path:
            cfg_out["hot_path_ms"] = {"train": hot_path["train"], "eval": hot_path["eval"]}
        # `roofline`: the contract's fields + the per-unit view; the long notes move to the detail file
        roof = {k: v for k, v in roofline.items() if k not in ("note", "traffic_note", "valu", "kernel_ms_min_median_max", "l1_accesses_per_launch")}
        line = {
            "metric": "rendered rays/sec (6-cam 450x800, 128 samples/ray)",
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if split else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfg_out, "roofline": roof, "cpu_baseline": cpu_baseline, "ranks_seen": ranks_seen,
        }
        if world > 1:   # the all-reduced quantity itself: mean rendered depth over every rank's rays (same on all ranks)
            # (slots hold depth SUMS over all ranks; scaled here, after every timed section: a first-use kernel load between
            # the timed loop and the event-timed launches idles the GPU long enough to drop its clocks, measured +10 %)
            line["allreduced_mean_depth_m"] = round(float(losses[args.warmup:].mean()) / rays_per_step_all_ranks, 4)
        if gpu_torch_baseline:
            line["gpu_torch_baseline"] = {k: gpu_torch_baseline[k] for k in ("value", "unit", "speedup_of_value")}
        if roofline_msda and "kernels" in roofline_msda:
            # [kernel, shape, ms, fraction of 8 TB/s on algorithmic bytes, fraction of the 39 TB/s vector-L1 gather ceiling]
            # the rows of the kernels the modules run (head-major value; all rows incl. the mmcv-layout ones: detail file)
            line["roofline_msda"] = {"bound": "hbm (second number: vector-L1 gathers, the operative bound)", "rows": [
                [r["kernel"][:40], r["shape"].split(":")[0], r["ms"], r["frac_of_8TBps"], r["l1_gather_frac"]]
                for r in roofline_msda["kernels"] if "head-major" in r["kernel"]]}
        elif roofline_msda:
            line["roofline_msda"] = roofline_msda
        if roofline_linear and "layer_sum" in roofline_linear:
            line["roofline_linear"] = {k: roofline_linear[k] for k in ("bound", "peak", "unit", "kernel", "matrix_ceiling_TFLOPs_f32_equiv",
                                                                       "layer_sum", "merged_vs_pairs_us") if k in roofline_linear}
            line["roofline_linear"]["frac_by_shape"] = {k: v["frac"] for k, v in roofline_linear["shapes"].items()}
        elif roofline_linear:
            line["roofline_linear"] = roofline_linear
        if roofline_bwd:
            line["roofline_bwd"] = {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "kernels"})
                                    for k, v in roofline_bwd.items()}
        if strong:
            line["strong_scaling"] = strong
        if extras:
            line["extras"] = extras
        # last: what the kept tail of the line should show
        if par_c:
            line["parity"] = par_c
        if hot_path:
            line["hot_path"] = hot_path
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as fd:
                json.dump({"line": line, "detail": detail}, fd, indent=1)
            line["detail_file"] = "gpurun_out/bench_detail.json"
        except OSError:
            pass
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
