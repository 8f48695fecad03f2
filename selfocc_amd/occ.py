"""Host side of the occupancy evaluation tail: dense field query,
Occ3D resample / threshold / LUT and the integer IoU counts.  Mirrors
NeuSHead.get_uniform_sdf (model/head/neus_head/neus_head.py:265-293), the tail of
eval_iou.py:211-250 and MeanIoU (utils/metric_util.py:66-165)."""
import numpy as np
import torch
import torch.distributed as dist

from . import abi
from ._lib import lib, check, ptr, current_stream

# utils/metric_util.py:37-64 (openseed2nuscenes): OpenSeeD class id -> nuScenes-lidarseg id
OPENSEED2NUSCENES = [1, 2, 3, 4, 5, 5, 6, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 15, 15, 16, 0]


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")


def field_query(vol, xyz, want_sdf=True, want_logits=False, want_argmax=False):
    """Trilinear lookup of an SDFVolume at metre positions xyz (n, 3).
    Returns dict(sdf (n), logits (n, n_sem), argmax (n) int32) for the requested parts."""
    _need_cuda(vol.sdf, "field_query")
    xyz = xyz.contiguous().float()
    n = xyz.shape[0]
    a = abi.SoQueryArgs()
    a.map = vol.mapping.to_abi()
    a.sdf_vol = ptr(vol.sdf)
    a.n_rgb, a.n_sem = vol.n_rgb, vol.n_sem
    if vol.feat is not None:
        a.feat_vol = ptr(vol.feat)
        a.feat_dtype = abi.DTYPE_F32 if vol.feat.dtype == torch.float32 else abi.DTYPE_BF16
        a.feat_stride = vol.feat.shape[3]
    a.xyz, a.n = ptr(xyz), n
    out = {}
    if want_sdf:
        out['sdf'] = torch.empty(n, device=xyz.device)
        a.sdf = ptr(out['sdf'])
    if want_logits:
        out['logits'] = torch.empty(n, vol.n_sem, device=xyz.device)
        a.sem_logits = ptr(out['logits'])
    if want_argmax:
        out['argmax'] = torch.empty(n, dtype=torch.int32, device=xyz.device)
        a.sem_argmax = ptr(out['argmax'])
    check(lib().selfocc_field_query(a, current_stream(xyz.device)), "selfocc_field_query")
    return out


class FieldQueryFunction(torch.autograd.Function):
    """Differentiable trilinear query of the field volume at fixed metre positions: ``sdf (n)`` (and the
    semantic ``logits (n, n_sem)``) attached to the autograd graph of ``sdf_vol`` / ``feat_vol``.  The reference
    gets this gradient from autograd through ``F.grid_sample`` inside ``get_uniform_sdf``
    (model/head/neus_head/neus_head.py:265-293, 532-538) — it is what makes ``uniform_sdf`` trainable for
    SoftSparsityLoss / SparsityLoss (loss/sparsity_loss.py:28-81; config/kitti/kitti_occ.py:135-138,
    config/nuscenes/nuscenes_occ_bev.py:157-160).  Positions carry no gradient (a fixed lattice)."""

    @staticmethod
    def forward(ctx, sdf_vol, feat_vol, xyz, mapping, n_rgb, n_sem, want_logits):
        vol_feat = feat_vol if (feat_vol is not None and feat_vol.numel() > 0) else None
        from .render import SDFVolume
        vol = SDFVolume(mapping, sdf_vol.detach(), None if vol_feat is None else vol_feat.detach(), n_rgb, n_sem)
        q = field_query(vol, xyz, want_sdf=True, want_logits=want_logits)
        ctx.save_for_backward(sdf_vol, feat_vol if vol_feat is not None else sdf_vol.new_zeros(0), xyz)
        ctx.meta = (mapping, n_rgb, n_sem, want_logits, vol_feat is not None)
        if want_logits:
            return q['sdf'], q['logits']
        return q['sdf'], sdf_vol.new_zeros(0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_logits):
        sdf_vol, feat_vol, xyz = ctx.saved_tensors
        mapping, n_rgb, n_sem, want_logits, has_feat = ctx.meta
        a = abi.SoQueryArgs()
        a.map = mapping.to_abi()
        a.sdf_vol = ptr(sdf_vol)
        a.n_rgb, a.n_sem = n_rgb, n_sem
        xyz = xyz.contiguous().float()
        a.xyz, a.n = ptr(xyz), xyz.shape[0]
        g_sdf_vol = g_feat_vol = None
        gs = gl = None
        if ctx.needs_input_grad[0] and g_sdf is not None:
            gs = g_sdf.contiguous().float()
            g_sdf_vol = torch.zeros_like(sdf_vol, dtype=torch.float32)
        if want_logits and has_feat and ctx.needs_input_grad[1] and g_logits is not None and g_logits.numel() > 0:
            if feat_vol.dtype != torch.float32:
                raise NotImplementedError("gradient of the semantic query needs a float32 feature volume")
            a.feat_vol, a.feat_dtype, a.feat_stride = ptr(feat_vol), abi.DTYPE_F32, feat_vol.shape[3]
            gl = g_logits.contiguous().float()
            g_feat_vol = torch.zeros_like(feat_vol)
        if gs is not None or gl is not None:
            check(lib().selfocc_field_query_bwd(a, ptr(gs), ptr(gl), ptr(g_sdf_vol), ptr(g_feat_vol),
                                                current_stream(xyz.device)), "selfocc_field_query_bwd")
        return g_sdf_vol, g_feat_vol, None, None, None, None, None


def field_query_autograd(vol, xyz, want_logits=False):
    """``field_query`` under autograd: dict(sdf (n)[, logits (n, n_sem)]) differentiable w.r.t. ``vol.sdf`` /
    ``vol.feat``."""
    _need_cuda(vol.sdf, "field_query_autograd")
    feat = vol.feat if (vol.feat is not None and want_logits) else vol.sdf.new_zeros(0)
    sdf, logits = FieldQueryFunction.apply(vol.sdf, feat, xyz.reshape(-1, 3).detach(), vol.mapping, vol.n_rgb, vol.n_sem,
                                           bool(want_logits and vol.n_sem > 0))
    out = {'sdf': sdf}
    if want_logits and vol.n_sem > 0:
        out['logits'] = logits
    return out


_LATTICES = {}
_LUTS = {}


def uniform_lattice(aabb, resolution, device, shift=False):
    """xyz lattice of NeuSHead.get_uniform_sdf (neus_head.py:266-281): (H, W, D, 3).  The unshifted lattice is a constant of
    (aabb, resolution, device): built once (it was three linspace kernels, a stack and a 7.7 MB copy per evaluation frame).
    The cached tensor is handed to every caller (NeuSHead.forward_occ returns it as `xyz`), so it is READ-ONLY by
    contract; an in-place edit by a caller bumps the tensor's version counter, which is checked here: an edited lattice is
    rebuilt, never served again."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())      # 'cuda' and 'cuda:0' are one cache entry
    key = (tuple(float(v) for v in aabb), float(resolution), dev.type, dev.index)
    if not shift and key in _LATTICES:
        xyz, version = _LATTICES[key]
        if xyz._version == version:
            return xyz
        del _LATTICES[key]
    xs = torch.linspace(aabb[0], aabb[3], int((aabb[3] - aabb[0]) / resolution), device=dev)
    ys = torch.linspace(aabb[1], aabb[4], int((aabb[4] - aabb[1]) / resolution), device=dev)
    zs = torch.linspace(aabb[2], aabb[5], int((aabb[5] - aabb[2]) / resolution), device=dev)
    W, H, D = len(xs), len(ys), len(zs)
    xyz = torch.stack([xs[None, :, None].expand(H, W, D), ys[:, None, None].expand(H, W, D),
                       zs[None, None, :].expand(H, W, D)], dim=-1)
    if shift:
        return xyz + torch.rand_like(xyz) * resolution
    if len(_LATTICES) > 8:
        _LATTICES.clear()
    _LATTICES[key] = (xyz, xyz._version)
    return xyz


def occ_resample(grid, coords, thresh, *, logits=None, lut=None, crop=(0, 0, 0, 0, 0, 0), density=False,
                 want_sampled=False):
    """grid (H, W, D) f32; coords (n0, n1, n2, 3) normalised [0,1] along (H, W, D);
    logits (H, W, D, C) or None.  Returns dict(occ int32 (n0,n1,n2)[, sem, sampled])."""
    _need_cuda(grid, "occ_resample")
    n0, n1, n2, _ = coords.shape
    a = abi.SoOccArgs()
    grid = grid.contiguous().float()
    coords = coords.contiguous().float()
    a.grid, a.coords = ptr(grid), ptr(coords)
    a.H, a.W, a.D = grid.shape
    a.n0, a.n1, a.n2 = n0, n1, n2
    for i in range(6):
        a.crop[i] = int(crop[i])
    a.thresh, a.density = float(thresh), int(bool(density))
    out = {'occ': torch.empty(n0, n1, n2, dtype=torch.int32, device=grid.device)}
    a.occ = ptr(out['occ'])
    keep = []
    if logits is not None:
        logits = logits.contiguous().float()
        a.logits, a.C = ptr(logits), logits.shape[3]
        if lut is not None:
            if torch.is_tensor(lut):
                lut_t = lut.to(device=grid.device, dtype=torch.int32)
            else:      # a python list (OPENSEED2NUSCENES): uploaded once per device, not once per frame (a blocking pageable copy)
                key = (tuple(int(v) for v in lut), str(grid.device))
                lut_t = _LUTS.get(key)
                if lut_t is None:
                    from ._lib import upload
                    lut_t = _LUTS[key] = upload(np.asarray(key[0], dtype=np.int32), grid.device, torch.int32)
            assert lut_t.numel() == a.C
            a.lut = ptr(lut_t)
            keep.append(lut_t)
        out['sem'] = torch.empty(n0, n1, n2, dtype=torch.int32, device=grid.device)
        a.sem = ptr(out['sem'])
    if want_sampled:
        out['sampled'] = torch.empty(n0, n1, n2, device=grid.device)
        a.sampled = ptr(out['sampled'])
    check(lib().selfocc_occ_resample(a, current_stream(grid.device)), "selfocc_occ_resample")
    return out


class MeanIoU:
    """Same interface and results as the reference's MeanIoU (utils/metric_util.py:66-165);
    the per-class python loop of ``.item()`` syncs (:113-121) is one kernel with integer
    atomics, the totals stay on the device until ``_after_epoch``."""

    def __init__(self, class_indices, empty_label, label_str, use_mask=False, dataset_empty_label=17,
                 name='none'):
        self.class_indices = list(class_indices)
        self.num_classes = len(self.class_indices)
        self.empty_label = empty_label
        self.dataset_empty_label = dataset_empty_label
        self.label_str = label_str
        self.use_mask = use_mask
        self.name = name
        self._cls = None

    def reset(self):
        dev = torch.device("cuda", torch.cuda.current_device())
        self.counts = torch.zeros(3, self.num_classes + 1, dtype=torch.int64, device=dev)
        self._cls = torch.tensor(self.class_indices, dtype=torch.int32, device=dev)

    @property
    def total_seen(self):
        return self.counts[0].float()

    @property
    def total_correct(self):
        return self.counts[1].float()

    @property
    def total_positive(self):
        return self.counts[2].float()

    def _after_step(self, outputs, targets, mask=None):
        if not isinstance(targets, (torch.Tensor, np.ndarray)):
            assert mask is None
            labels = torch.from_numpy(targets['semantics']).cuda()
            masks = torch.from_numpy(targets['mask_camera']).bool().cuda()
            targets = labels
            targets[targets == self.dataset_empty_label] = self.empty_label
            nz = (targets != self.empty_label).nonzero()[:, 2]
            outputs[..., (nz.max() + 1):] = self.empty_label
            outputs[..., :nz.min()] = self.empty_label
            mask = masks if self.use_mask else None
        if isinstance(targets, np.ndarray):
            targets = torch.from_numpy(targets).to(outputs.device)
        _need_cuda(outputs, "MeanIoU")
        p = outputs.reshape(-1).to(torch.int32).contiguous()
        t = targets.reshape(-1).to(torch.int32).contiguous()
        m = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()
        check(lib().selfocc_iou_counts(ptr(p), ptr(t), ptr(m), p.numel(), ptr(self._cls), self.num_classes,
                                       int(self.empty_label), ptr(self.counts), current_stream(p.device)),
              "selfocc_iou_counts")

    def _after_epoch(self):
        if dist.is_initialized():
            dist.all_reduce(self.counts)  # RCCL; integer sums are exact
            dist.barrier()
        seen, correct, positive = (self.counts[i].cpu().double() for i in range(3))
        ious, precs, recas = [], [], []
        for i in range(self.num_classes):
            precs.append(0. if positive[i] == 0 else (correct[i] / positive[i]).item())
            if seen[i] == 0:
                ious.append(1); recas.append(1)
            else:
                ious.append((correct[i] / (seen[i] + positive[i] - correct[i])).item())
                recas.append((correct[i] / seen[i]).item())
        miou = np.mean(ious) if ious else 0.0
        self.per_class = dict(iou=ious, precision=precs, recall=recas)
        occ_iou = (correct[-1] / (seen[-1] + positive[-1] - correct[-1])).item()
        return miou * 100, occ_iou * 100
