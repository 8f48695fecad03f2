"""Diagnostic (GPU): ONE process, two HIP streams.  Stream B replays the msda ops captured from a shipped-size eval encoder
pass and compares every result bitwise with the first; stream A runs a disturber at the same time (whole encoder passes, or
one captured op in a loop).  Tells a cross-PROCESS effect (nothing here) from a concurrent-KERNEL effect (reproduces here)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
d = torch.device("cuda:0")
import hotpath_common as hc
import selfocc_amd.model.bricks as bricks
import selfocc_amd.model.encoder.attention as attention
import selfocc_amd.model.encoder.tpvformer as tpvformer

def clone(x):
    if torch.is_tensor(x):
        return x.detach().clone()
    if isinstance(x, (list, tuple)):
        return type(x)(clone(y) for y in x)
    if isinstance(x, dict):
        return {k: clone(v) for k, v in x.items()}
    return x

def sig(x):
    if torch.is_tensor(x):
        return (tuple(x.shape), str(x.dtype), x.stride())
    if isinstance(x, (list, tuple)):
        return tuple(sig(y) for y in x)
    if isinstance(x, dict):
        return tuple((k, sig(v)) for k, v in sorted(x.items()))
    if isinstance(x, torch.nn.Module):
        return type(x).__name__
    return x if isinstance(x, (int, float, bool, str, type(None))) else type(x).__name__

calls = {}
capturing = [True]
def rec(name, fn):
    def w(*a, **k):
        if capturing[0]:
            key = (name, sig(a), sig(k))
            if key not in calls:
                calls[key] = (fn, clone(a), clone(k))
        return fn(*a, **k)
    return w
for mod in (bricks, attention, tpvformer):
    for name in ("linear_fwd", "linear_fwd_heads", "msda_fused_inference", "msda_cross_inference", "fused_linear",
                 "value_proj_head_major", "point_sampling"):
        if hasattr(mod, name):
            setattr(mod, name, rec(f"{mod.__name__.split('.')[-1]}.{name}", getattr(mod, name)))
torch.manual_seed(0)
cfg = hc.shipped("nuscenes_occ")
lifter, enc, _h, _ = hc.build(cfg, d)
enc.eval()
img = tuple(cfg['img_size'])
c2w, l2i, K = hc.ring_cameras(6, img, 1266.0)
metas = [dict(lidar2img=l2i, img2lidar=c2w, img_shape=img)]
g = torch.Generator().manual_seed(5)
feats = [torch.randn(1, 6, 96, -(-img[0] // s_), -(-img[1] // s_), generator=g).to(d) for s_ in (8, 16, 32, 64)]
enc.layers = enc.layers[:1]
def flat(x):
    if torch.is_tensor(x):
        return [x]
    if isinstance(x, (list, tuple)):
        return [t for y in x for t in flat(y)]
    return []
with torch.no_grad():
    enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)
    torch.cuda.synchronize()
    capturing[0] = False
    victims = [(n, v) for (n, _, _), v in calls.items() if "msda" in n]
    if os.environ.get("DIAG_PLAIN_ONLY"):
        victims = []
    if os.environ.get("DIAG_EXTRA_VICTIMS"):
        from selfocc_amd.msda import multi_scale_deformable_attn
        gg = torch.Generator(device=d).manual_seed(3)
        shp = torch.tensor([[116, 200], [58, 100], [29, 50]], device=d)
        lsi = torch.tensor([0, 23200, 29000], device=d)
        val = torch.randn(1, 30450, 6, 16, device=d, generator=gg)
        loc = torch.rand(1, 66049, 6, 3, 8, 2, device=d, generator=gg) * 1.1 - 0.05
        aw = torch.softmax(torch.randn(1, 66049, 6, 24, device=d, generator=gg), -1).view(1, 66049, 6, 3, 8)
        victims.append(("plain msda_fwd (sampling_locations form)", (multi_scale_deformable_attn, (val, shp, lsi, loc, aw), {})))
    if os.environ.get("DIAG_EXTRA_VICTIMS") and not os.environ.get("DIAG_PLAIN_ONLY"):
        im = torch.randn(6, 16, 116, 200, device=d, generator=gg)
        grid = torch.rand(6, 400, 400, 2, device=d, generator=gg) * 2.2 - 1.1
        victims.append(("torch grid_sample", (torch.nn.functional.grid_sample, (im, grid), dict(align_corners=False))))
        xa = torch.randn(78899, 96, device=d, generator=gg)
        victims.append(("torch elementwise + row sum", ((lambda t: (t * 1.5 + 0.25).tanh().sum(-1)), (xa,), {})))
        ia = torch.randint(0, 78899, (400000,), device=d, generator=gg)
        victims.append(("torch index_select (row gather)", ((lambda t, i: t.index_select(0, i)), (xa, ia), {})))
    ln = torch.nn.LayerNorm(96).to(d)
    xx = torch.randn(78899, 96, device=d)
    mm = torch.randn(4096, 4096, device=d)
    big = torch.randn(1 << 28, device=d)          # 1 GiB
    gidx = torch.randint(0, 1 << 22, (1 << 24,), device=d)
    gtab = torch.randn(1 << 22, 16, device=d)
    disturbers = {"none": None, "torch_matmul": lambda: mm @ mm, "hbm_copy": lambda: big.clone(),
                  "torch_gather": lambda: gtab.index_select(0, gidx), "encoder_pass": lambda: enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas),
                  "lifter": lambda: lifter(feats), "torch_elementwise": lambda: (xx * 2 + 1).relu().sum(),
                  "layernorm": lambda: enc.layers[0].norms[0](xx[None]) if hasattr(enc.layers[0], 'norms') else ln(xx)}
    for (n, sa_, sk_), (fn, a, k) in calls.items():
        if "msda" not in n:
            if n not in disturbers and os.environ.get("DIAG_SHOW_SIG"):
                print("disturber signature", n, sa_, sk_, flush=True)
            disturbers.setdefault(n, (lambda fn=fn, a=a, k=k: fn(*a, **k)))
    want = os.environ.get("DIAG_DISTURB", "none,encoder_pass").split(",")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    import ctypes as C
    probe = C.CDLL(os.path.join(ROOT, "scripts", "micro", "libxlane_probe.so"))
    sink = torch.zeros(16, device=d)
    px, py = torch.randn(78899 // 16 * 16, 96, device=d), torch.empty(78899 // 16 * 16, 96, device=d)
    for kind, nm, iters in ((1, "micro:mfma_bf16", 40000), (2, "micro:mfma_f32", 20000), (3, "micro:cvt", 60000), (4, "micro:lds", 60000),
                            (5, "micro:b3like", 1), (5, "micro:b3like-noMFMA", 1 | (1 << 8)), (5, "micro:b3like-noLDS", 1 | (2 << 8)),
                            (5, "micro:b3like-noStore", 1 | (4 << 8)), (5, "micro:b3like-noCvt", 1 | (8 << 8)),
                            (5, "micro:b3like-noLoad", 1 | (16 << 8)), (5, "micro:b3like-onlyMFMA", 1 | (30 << 8)),
                            (5, "micro:b3like-onlyStore", 1 | (27 << 8)), (5, "micro:b3like-onlyLoad", 1 | (15 << 8))):
        disturbers[nm] = (lambda kind=kind, iters=iters: probe.probe_disturber(
            kind, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(sink.data_ptr()), iters,
            C.c_void_p(px.data_ptr()), C.c_void_p(py.data_ptr()), px.shape[0]))
    if os.environ.get("DIAG_MICRO_VICTIMS"):
        table = torch.empty(1 << 22, 4, dtype=torch.int32, device=d)
        probe.probe_fill_table(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(table.data_ptr()), 1 << 22)
        torch.cuda.synchronize()
        names = ["dpp", "ds_bpermute/permute", "gather dwordx4", "exp/rcp", "v_pk_fma_f32", "f32 division", "64-bit address math", "16 gathers in flight", "32 gathers in flight", "pk_mul swizzled fresh-cvt", "pk_mul plain fresh-cvt",
                 "pk_mul swizzled old regs", "pk_fma broadcast", "pk_add inline const", "pk_fma op_sel_hi:[1,0,1]",
                 "pk_fma op_sel_hi:[0,1,0] const", "pk_fma op_sel_hi:[1,0,0] const"]
        for dn in want:
            cnt = torch.zeros(24, dtype=torch.int64, device=d)
            for it in range(int(os.environ.get("DIAG_REPEAT", "200"))):
                if disturbers[dn] is not None:
                    with torch.cuda.stream(sa):
                        disturbers[dn]()
                with torch.cuda.stream(sb):
                    probe.probe_victims(C.c_void_p(sb.cuda_stream), C.c_void_p(cnt.data_ptr()), it * 7919, C.c_void_p(table.data_ptr()), 1 << 20)
            torch.cuda.synchronize()
            print(f"disturber {dn:28s} micro victims, wrong results:", dict(zip(names, cnt[:17].tolist())), flush=True)
        sys.exit(0)
    n = int(os.environ.get("DIAG_REPEAT", "200"))
    for dn in want:
        dist = disturbers[dn]
        for vn, (fn, a, k) in victims:
            first, bad = None, 0
            shown = [0]
            backup = clone((a, k))
            with torch.cuda.stream(sb):
                quiet = [t.clone() for t in flat(fn(*a, **k))]          # before the disturber starts
            torch.cuda.synchronize()
            for it in range(n):
                if dist is not None:
                    with torch.cuda.stream(sa):
                        dist()
                with torch.cuda.stream(sb):
                    out = [t.clone() for t in flat(fn(*a, **k))]
                    if first is None:
                        first = out
                    elif any(not torch.equal(x, y) for x, y in zip(first, out)):
                        bad += 1
                    if os.environ.get("DIAG_PATTERN") and not torch.equal(out[0], quiet[0]) and shown[0] < 6:
                        shown[0] += 1
                        w, r = out[0].reshape(-1, 16), quiet[0].reshape(-1, 16)
                        rows = (w != r).any(-1).nonzero().flatten()
                        print(f"   [{vn}] replay {it}: {rows.numel()} wrong (query, head) rows: {rows[:16].tolist()}", flush=True)
                        for rr in rows[:3].tolist():
                            print(f"      row {rr} (q {rr // 6}, h {rr % 6}) channels wrong {int((w[rr] != r[rr]).sum())}: wrong - right = "
                                  f"{[round(v, 4) for v in (w[rr] - r[rr]).tolist()]}  right = {[round(v, 4) for v in r[rr].tolist()]}", flush=True)
            torch.cuda.synchronize()
            after = [t.clone() for t in flat(fn(*a, **k))]                # after it stopped
            torch.cuda.synchronize()
            stomped = [i for i, (x, y) in enumerate(zip(flat(backup), flat((a, k)))) if not torch.equal(x, y)]
            from selfocc_amd._lib import lib as _solib
            if hasattr(_solib(), "selfocc_diag_read"):
                import ctypes as C
                h8 = (C.c_uint32 * (8 + 8 * 24))()
                _solib().selfocc_diag_read(h8, 1)
                print(f"      redundant-evaluation disagreements (lanes): scalar state / gather / reduction {h8[0]}, point record {h8[1]}: "
                      f"aw {h8[2]}, w {h8[3]}, off {h8[4]}", flush=True)
                import struct
                fl = lambda u: struct.unpack('f', struct.pack('I', u))[0]
                for sl in range(min(8, h8[1])):
                    r = h8[8 + 24 * sl: 8 + 24 * sl + 24]
                    print(f"         gid {r[0]} thread {r[1]} point {r[2]} block {r[3]}: off {[int(v) for v in r[4:8]]} vs {[int(v) for v in r[8:12]]}; "
                          f"w {[round(fl(v), 5) for v in r[12:16]]} vs {[round(fl(v), 5) for v in r[16:20]]}; aw {fl(r[20]):.5f} vs {fl(r[21]):.5f}", flush=True)
            print(f"disturber {dn:28s} victim {vn:36s} non-repeatable {bad}/{n - 1}   first == quiet run: "
                  f"{all(torch.equal(x, y) for x, y in zip(first, quiet))}   quiet run after == before: "
                  f"{all(torch.equal(x, y) for x, y in zip(after, quiet))}   victim inputs changed: {stomped}", flush=True)
