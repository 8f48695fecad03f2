"""The hot-path part of the reference's SHIPPED experiment configs as JSON (scripts/shipped_cfg/*.json), so that the
GPU box — where /root/reference does not exist — times exactly what config/**/*.py describes instead of a hand-restated dict.
Read through selfocc_amd.config.Config.fromfile (the drop-in loader: `_base_` inheritance, `_delete_`), unchanged values:
    model.lifter / model.encoder / model.head, loss, loss_input_convertion, img_size[, crop_size], num_rays, optimizer,
    grad_max_norm, amp
(the image backbone / neck, datasets and schedules are out of scope).  Run where the reference is mounted:
    python scripts/dump_shipped_configs.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from selfocc_amd.config import Config

REF = os.environ.get("SELFOCC_REFERENCE", "/root/reference")
# every experiment config the reference ships (config/{nuscenes,kitti,kitti_raw}/*.py), not a selection
WANT = {"nuscenes_occ": "config/nuscenes/nuscenes_occ.py", "nuscenes_occ_bev": "config/nuscenes/nuscenes_occ_bev.py",
        "nuscenes_depth": "config/nuscenes/nuscenes_depth.py", "nuscenes_novel_depth": "config/nuscenes/nuscenes_novel_depth.py",
        "kitti_occ": "config/kitti/kitti_occ.py", "kitti_novel_depth": "config/kitti/kitti_novel_depth.py",
        "kitti_raw_depth": "config/kitti_raw/kitti_raw_depth.py"}
import glob
assert sorted(WANT.values()) == sorted(os.path.relpath(p, REF) for d in ("nuscenes", "kitti", "kitti_raw")
                                       for p in glob.glob(os.path.join(REF, "config", d, "*.py"))), "a shipped config is not listed"
for name, rel in WANT.items():
    c = Config.fromfile(os.path.join(REF, rel)).to_dict()
    out = {"source": rel,
           "model": {k: c["model"][k] for k in ("type", "lifter", "encoder", "head")},
           "loss": c["loss"], "loss_input_convertion": c["loss_input_convertion"],
           "img_size": c["img_size"], "num_rays": c["num_rays"], "optimizer": c["optimizer"],
           "grad_max_norm": c["grad_max_norm"], "amp": c.get("amp", False)}
    if "crop_size" in c:
        out["crop_size"] = c["crop_size"]
    path = os.path.join(ROOT, "scripts", "shipped_cfg", name + ".json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")
