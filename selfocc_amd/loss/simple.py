"""The small reductions of the loss layer, torch on per-ray tensors.

RGBLossMS / SemLossMS / SemCELossMS <- loss/rgb_loss_ms.py:41-213
EikonalLoss <- loss/eikonal_loss.py:6-21      SecondGradLoss <- loss/second_grad_loss.py:6-19
EdgeLoss3DMS <- loss/edge_loss_3d_ms.py:8-79  sparsity family <- loss/sparsity_loss.py:6-113
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..registry import OPENOCC_LOSS
from .base import BaseLoss
from .reproj import SSIM


def _lattice_grid(rays, img_size, n):
    g = rays.reshape(1, 1, -1, 2).repeat(n, 1, 1, 1).clone()
    g[..., 0] /= img_size[1]
    g[..., 1] /= img_size[0]
    return g * 2 - 1


@OPENOCC_LOSS.register_module()
class RGBLossMS(BaseLoss):
    def __init__(self, weight=1.0, img_size=None, no_ssim=True, ray_resize=None, input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'ms_colors': 'ms_colors', 'ms_rays': 'ms_rays', 'gt_imgs': 'gt_imgs'}
        assert img_size is not None
        self.img_size, self.ray_resize = img_size, ray_resize
        self.no_ssim = no_ssim or ray_resize is None
        if not self.no_ssim:
            self.ssim = SSIM()
        self.loss_func = self.rgb_loss

    def rgb_loss(self, ms_colors, ms_rays, gt_imgs):
        bs, num_cams = gt_imgs.shape[:2]
        if isinstance(ms_rays, list):
            raise NotImplementedError
        gt = F.grid_sample(gt_imgs.flatten(0, 1), _lattice_grid(ms_rays, self.img_size, bs * num_cams),
                           mode='bilinear', padding_mode='zeros', align_corners=True)
        gt = gt.reshape(bs, num_cams, 3, -1).transpose(-1, -2)
        tot = 0.
        for color in ms_colors:
            loss = torch.abs(color - gt).mean()
            if not self.no_ssim:
                im = lambda t: t.reshape(bs * num_cams, *self.ray_resize, 3).permute(0, 3, 1, 2)
                loss = 0.15 * loss + 0.85 * self.ssim(im(color), im(gt)).mean()
            tot = tot + loss
        return tot / len(ms_colors)


class _SemBase(BaseLoss):
    def __init__(self, weight=1.0, img_size=None, ray_resize=None, input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'sem': 'sem', 'metas': 'metas', 'ms_rays': 'ms_rays'}
        assert img_size is not None
        self.img_size, self.ray_resize = img_size, ray_resize
        self.loss_func = self.sem_loss

    def _one_hot_gt(self, sem, metas, ms_rays):
        gt_imgs = [m['sem'] for m in metas]
        if isinstance(gt_imgs[0], np.ndarray):
            gt_imgs = sem[0].new_tensor(np.asarray(gt_imgs), dtype=torch.long)
        elif isinstance(gt_imgs[0], torch.Tensor):
            gt_imgs = torch.stack(gt_imgs).to(sem[0].device)
        else:
            raise NotImplementedError
        if isinstance(ms_rays, list):
            raise NotImplementedError
        rays = ms_rays.to(torch.long)
        gt = gt_imgs[:, :, rays[:, 1], rays[:, 0]]
        return F.one_hot(gt, num_classes=sem[0].shape[-1]).to(torch.float)


@OPENOCC_LOSS.register_module()
class SemLossMS(_SemBase):
    def sem_loss(self, sem, metas, ms_rays):
        gt = self._one_hot_gt(sem, metas, ms_rays)
        return sum(F.binary_cross_entropy(torch.clamp(s, 0, 1), gt) for s in sem) / len(sem)


@OPENOCC_LOSS.register_module()
class SemCELossMS(_SemBase):
    def sem_loss(self, sem, metas, ms_rays):
        gt = self._one_hot_gt(sem, metas, ms_rays)
        return sum(torch.mean(torch.sum(-torch.log(torch.clamp(s, 1e-6, 1)) * gt, dim=-1)) for s in sem) / len(sem)


@OPENOCC_LOSS.register_module()
class EikonalLoss(BaseLoss):
    def __init__(self, weight=1.0, input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'eik_grad': 'eik_grad'}
        self.loss_func = self.eikonal

    supports_ray_shard = True

    def eikonal(self, eik_grad):
        from ..dist import shard_of, global_value_local_grad
        shard = getattr(self, '_ray_shard', None) or shard_of(eik_grad)     # explicit (outputs['ray_shard']) first
        n_local = eik_grad.numel() // max(eik_grad.shape[-1], 1)
        if (eik_grad.is_cuda and eik_grad.dtype == torch.float32 and eik_grad.shape[-1] == 3 and n_local > 0
                and not torch.is_autocast_enabled()):
            sq_sum = _EikonalSum.apply(eik_grad)           # one HIP pass per direction (csrc/eikonal.hip)
        else:
            sq_sum = ((eik_grad.norm(2, dim=-1) - 1) ** 2).sum()
        if shard is None:
            return sq_sum / n_local
        # ray-sharded head: this rank's samples only; the mean runs over the samples of ALL ranks
        n_cams = shard.full.img2lidar.shape[0]
        n_global = n_local // (n_cams * shard.rays_per_cam_local) * n_cams * shard.rays_per_cam_full
        return global_value_local_grad(sq_sum / n_global)


class _EikonalSum(torch.autograd.Function):
    """sum_i (||g_i||_2 - 1)^2 over the rows of (..., 3) through selfocc_eikonal_fwd / _bwd: torch ran a norm reduction over
    a dimension of three, sub, pow, sum forward and four more elementwise kernels backward (0.26 ms per iteration on
    7.4 M samples)."""

    @staticmethod
    def forward(ctx, g):
        from .._lib import lib, check, ptr, current_stream
        g2 = g.reshape(-1, 3).contiguous()
        n = g2.shape[0]
        part = torch.empty(int(lib().selfocc_eikonal_partials(n)), device=g.device, dtype=torch.float32)
        check(lib().selfocc_eikonal_fwd(ptr(g2), ptr(part), n, current_stream(g.device)), "selfocc_eikonal_fwd")
        ctx.save_for_backward(g2)
        ctx.shape = g.shape
        return part.sum()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        from .._lib import lib, check, ptr, current_stream
        (g2,) = ctx.saved_tensors
        gg = torch.empty_like(g2)
        scale = go.reshape(1).float().contiguous()
        check(lib().selfocc_eikonal_bwd(ptr(g2), ptr(scale), ptr(gg), g2.shape[0], current_stream(g2.device)),
              "selfocc_eikonal_bwd")
        return gg.view(ctx.shape)


@OPENOCC_LOSS.register_module()
class SecondGradLoss(BaseLoss):
    def __init__(self, weight=1.0, input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'second_grad': 'second_grad'}
        self.loss_func = lambda second_grad: second_grad.abs().mean()


def get_smooth_loss(disp, img):
    """edge-aware first-order smoothness of a (normalised) depth lattice"""
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


@OPENOCC_LOSS.register_module()
class EdgeLoss3DMS(BaseLoss):
    def __init__(self, weight=1.0, input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'curr_imgs': 'curr_imgs', 'ms_depths': 'ms_depths', 'ms_rays': 'ms_rays'}
        self.img_size = kwargs.get('img_size', [768, 1600])
        self.ray_resize = kwargs.get('ray_resize', None)
        self.use_inf_mask = kwargs.get('use_inf_mask', False)
        assert self.ray_resize is not None
        self.loss_func = self.edge_loss

    def edge_loss(self, curr_imgs, ms_depths, ms_rays, ms_accs=None, max_depths=None):
        if self.use_inf_mask:
            assert ms_accs is not None and max_depths is not None
        if not isinstance(ms_rays, list):
            ms_rays = [ms_rays] * len(ms_depths)
        bs, num_cams, num_rays = ms_depths[0].shape
        tot = 0.
        for scale, (depth, rays) in enumerate(zip(ms_depths, ms_rays)):
            rgb = F.grid_sample(curr_imgs.flatten(0, 1), _lattice_grid(rays, self.img_size, bs * num_cams),
                                mode='bilinear', padding_mode='border', align_corners=True)
            rgb = rgb.reshape(bs * num_cams, -1, *self.ray_resize)
            if self.use_inf_mask:
                depth = depth * ms_accs[scale] + max_depths[scale] * (1 - ms_accs[scale])
            depth = depth.reshape(bs * num_cams, 1, *self.ray_resize)
            norm_depth = depth / (depth.mean(2, True).mean(3, True) + 1e-6)
            tot = tot + get_smooth_loss(norm_depth, rgb)
        return tot / len(ms_depths)


@OPENOCC_LOSS.register_module()
class SparsityLoss(BaseLoss):
    def __init__(self, weight=1.0, scale=1.0, input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'density': 'density'}
        self.scale = scale
        self.loss_func = lambda density: torch.pow(1.0 / torch.cosh(density / (2.0 * self.scale)), 2).mean()


@OPENOCC_LOSS.register_module()
class HardSparsityLoss(BaseLoss):
    def __init__(self, weight=1.0, scale=1.0, thresh=0.2, crop=[[0, 0], [0, 0], [0, 0]], input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'density': 'density'}
        self.scale, self.thresh, self.crop = scale, thresh, np.asarray(crop)
        self.loss_func = self.hard_sparsity_loss

    def hard_sparsity_loss(self, density):
        c = self.crop
        for ax in range(3):
            if c[ax, 0] > 0:
                density.narrow(ax, 0, int(c[ax, 0])).fill_(100)
            if c[ax, 1] > 0:
                density.narrow(ax, density.shape[ax] - int(c[ax, 1]), int(c[ax, 1])).fill_(100)
        return torch.relu(torch.sigmoid(-self.scale * density).mean() - self.thresh)


@OPENOCC_LOSS.register_module()
class SoftSparsityLoss(BaseLoss):
    def __init__(self, weight=1.0, input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'density': 'density'}
        self.loss_func = lambda density: torch.relu(-1 * density).mean()


@OPENOCC_LOSS.register_module()
class AdaptiveSparsityLoss(BaseLoss):
    def __init__(self, weight=1, input_dict=None, slack=4.0, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict or {'sdfs': 'sdfs', 'ts': 'ts', 'ms_depths': 'ms_depths'}
        self.slack = slack
        self.loss_func = self.adaptive_sparsity_loss

    def adaptive_sparsity_loss(self, sdfs, ts, ms_depths):
        depths = ms_depths[0]
        bs, num_cams, num_rays = depths.shape
        assert bs == 1
        ts = torch.stack(ts, dim=0).reshape(bs, num_cams, num_rays, -1)
        sdfs = torch.stack(sdfs, dim=0).reshape(bs, num_cams, num_rays, -1)
        return torch.relu(-1 * sdfs[ts > (depths + self.slack).unsqueeze(-1)]).mean()
