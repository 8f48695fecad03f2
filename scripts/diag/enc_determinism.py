"""Diagnostic (GPU): is the shipped-size eval encoder bitwise repeatable?  N forward passes in one process; per-layer-stage
monkeypatch counters not needed: compares the three output planes of every pass against the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import hotpath_common as hc
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
if os.environ.get("DIAG_POISON"):      # uninitialised-read probe: fill the caching allocator's pool with NaN before the passes
    junk = [torch.full((1 << 28,), float('nan'), device=d) for _ in range(8)]
    del junk
n = int(os.environ.get("DIAG_REPEAT", "10"))
n_layers = int(os.environ.get("DIAG_LAYERS", "4"))
enc.layers = enc.layers[:n_layers]
with torch.no_grad():
    first = None
    for it in range(n):
        out = [o.clone() for o in enc(lifter(feats)['representation'], ms_img_feats=feats, metas=metas)['representation']]
        if first is None:
            first = out
            continue
        rows = [int((a != b).any(-1).sum()) for a, b in zip(first, out)]
        mx = [float((a - b).abs().max()) for a, b in zip(first, out)]
        if any(rows):
            print(os.environ.get("DIAG_TAG", ""), f"pass {it}: rows differing from pass 0 per plane {rows}, max abs diff {mx}", flush=True)
print(os.environ.get("DIAG_TAG", ""), "done", n, "passes; checksum", [float(o.double().sum()) for o in first])
