"""Temporal reprojection photometric losses.

ReprojLossMonoMultiNewCombine <- loss/reproj_loss_mono_multi_new_combine.py:41-247
ReprojLossMonoMultiNew        <- loss/reproj_loss_mono_multi_new.py:41-287
The per-sample chain (projection, bilinear fetch, masks, per-ray renormalised reductions)
is ONE fused HIP launch per camera (reproj.py -> csrc/reproj.hip); SSIM, the auto-mask
minimum and the means stay in torch on the (R, 3) lattice images.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import upload
from ..registry import OPENOCC_LOSS
from ..reproj import ReprojSampleFunction
from .base import BaseLoss


class _SSIMFunction(torch.autograd.Function):
    """SSIM map through ``selfocc_ssim_fwd / _bwd`` (csrc/ssim.hip); inputs of any strides, float32."""

    @staticmethod
    def forward(ctx, x, y):
        import ctypes
        from .._lib import lib, check, ptr, current_stream
        x, y = x.float(), y.float()
        N, Cc, H, W = x.shape
        assert y.shape == x.shape
        out = torch.empty(N, Cc, H, W, device=x.device, dtype=torch.float32)
        xs = (ctypes.c_int64 * 4)(*x.stride()); ys = (ctypes.c_int64 * 4)(*y.stride())
        check(lib().selfocc_ssim_fwd(ptr(x), ptr(y), ctypes.cast(xs, ctypes.c_void_p), ctypes.cast(ys, ctypes.c_void_p),
                                     N, Cc, H, W, ptr(out), current_stream(x.device)), "selfocc_ssim_fwd")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        import ctypes
        from .._lib import lib, check, ptr, current_stream
        x, y = ctx.saved_tensors
        N, Cc, H, W = x.shape
        g = g.contiguous().float()
        gx = torch.empty(N, Cc, H, W, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        gy = torch.empty(N, Cc, H, W, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None
        xs = (ctypes.c_int64 * 4)(*x.stride()); ys = (ctypes.c_int64 * 4)(*y.stride())
        check(lib().selfocc_ssim_bwd(ptr(x), ptr(y), ctypes.cast(xs, ctypes.c_void_p), ctypes.cast(ys, ctypes.c_void_p),
                                     N, Cc, H, W, ptr(g), ptr(gx), ptr(gy), current_stream(x.device)), "selfocc_ssim_bwd")
        return gx, gy


class SSIM(nn.Module):
    """(1 - SSIM) / 2 with 3x3 mean filters on reflection-padded inputs, clamped to [0, 1]."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2

    def forward(self, x, y):
        # one HIP launch per direction instead of ~40 / ~80 tiny torch kernels; no CPU fallback (selfocc_amd/_lib.py)
        if not x.is_cuda:
            raise RuntimeError("SSIM needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
        return _SSIMFunction.apply(x, y)


# a transform that maps every point behind the camera: disables one temporal frame in the kernel
_INVALID = torch.diag(torch.tensor([1.0, 1.0, -1.0, 1.0]))
_INVALID_DEV = {}


def _invalid_on(device):
    """the constant on ``device``, uploaded once (a per-call ``.to(device)`` of a pageable host tensor is a blocking copy:
    found by tests/test_shipped_configs_gpu.py under set_sync_debug_mode("error") on the KITTI configs)"""
    key = str(device)
    if key not in _INVALID_DEV:
        _INVALID_DEV[key] = upload(_INVALID.numpy(), device, torch.float32)
    return _INVALID_DEV[key]


class _ReprojBase(BaseLoss):
    supports_ray_shard = True

    def __init__(self, weight=1.0, input_dict=None, **kwargs):
        super().__init__(weight)
        self.input_dict = input_dict if input_dict is not None else {
            'curr_imgs': 'curr_imgs', 'prev_imgs': 'prev_imgs', 'next_imgs': 'next_imgs',
            'ray_indices': 'ray_indices', 'weights': 'weights', 'ts': 'ts', 'metas': 'metas', 'ms_rays': 'ms_rays'}
        self.img_size = kwargs.get('img_size', [768, 1600])
        self.ray_resize = kwargs.get('ray_resize', None)
        self.no_automask = kwargs.get('no_automask', False)
        self.dims = kwargs.get('dims', 3)
        self.no_ssim = kwargs.get('no_ssim', False) or (self.ray_resize is None)
        if not self.no_ssim:
            self.ssim = SSIM()
        self.loss_func = self.reproj_loss
        self.iter_counter = 0

    def _transforms(self, metas, key, like, num_cams):
        vals = [m[key] for m in metas]
        if isinstance(vals[0], (np.ndarray, list)):
            t = upload(np.asarray(vals), like.device, like.dtype)
        else:
            t = torch.stack(vals, dim=0).to(like)
        return t.reshape(-1, num_cams, 4, 4).float()

    @staticmethod
    def _local_rows(shard, per_ray_full):
        """rows of a full-lattice per-ray tensor (R, ...) of ONE camera that belong to this rank's block"""
        from ..dist import row_block
        r0, r1 = row_block(shard.full.ny, shard.rank, shard.world_size)
        return per_ray_full.reshape(shard.full.ny, shard.full.nx, *per_ray_full.shape[1:])[r0:r1].reshape(
            -1, *per_ray_full.shape[1:])

    @staticmethod
    def _gather_per_ray(shard, l1, comb, any_valid):
        """(N, R_local), (N, R_local, 3), (N, R_local) of this rank's rows -> the full-lattice (N, R) / (N, R, 3) terms:
        ONE all-gather of 5 floats per ray; its backward hands every rank the gradient of its own rows."""
        from ..dist import gather_rays_autograd
        n = l1.shape[0]
        pack = torch.cat([l1[..., None], comb, any_valid[..., None]], -1)               # (N, R_local, 5)
        full = gather_rays_autograd(pack.reshape(-1, 5), shard.full).reshape(n, -1, 5)   # lattice order: cams, rows, cols
        return full[..., 0], full[..., 1:4], full[..., 4]

    def _sample_lattice(self, pix, img, padding_mode):
        """bilinear sample of one camera image (3, H, W) at the ray pixels -> (R, 3)."""
        g = pix.clone().reshape(1, 1, -1, 2)
        g[..., 0] /= self.img_size[1]
        g[..., 1] /= self.img_size[0]
        out = F.grid_sample(img[None], 2 * g - 1, mode='bilinear', padding_mode=padding_mode, align_corners=True)
        return out.reshape(img.shape[0], -1).transpose(0, 1)

    def _photometric(self, pred, target):
        """0.85 SSIM + 0.15 L1 per ray between two (R, 3) lattice images."""
        l1 = torch.abs(target - pred).mean(-1)
        if self.no_ssim:
            return l1
        im = lambda t: t.reshape(1, *self.ray_resize, self.dims).permute(0, 3, 1, 2)
        return 0.85 * self.ssim(im(pred), im(target)).mean(1).flatten() + 0.15 * l1

    def _sample_lattice_all(self, pix, imgs, padding_mode):
        """all cameras at once: imgs (N, 3, H, W) at the (shared) ray pixels -> (N, R, 3)."""
        g = pix.clone().reshape(1, 1, -1, 2)
        g[..., 0] /= self.img_size[1]
        g[..., 1] /= self.img_size[0]
        out = F.grid_sample(imgs, (2 * g - 1).expand(imgs.shape[0], -1, -1, -1), mode='bilinear',
                            padding_mode=padding_mode, align_corners=True)
        return out.reshape(imgs.shape[0], imgs.shape[1], -1).transpose(1, 2)

    def _photometric_all(self, pred, target, im):
        """batched _photometric: (N, R, 3) x (N, R, 3) -> (N, R)."""
        l1 = torch.abs(target - pred).mean(-1)
        if self.no_ssim:
            return l1
        return 0.85 * self.ssim(im(pred), im(target)).mean(1).flatten(1) + 0.15 * l1


@OPENOCC_LOSS.register_module()
class ReprojLossMonoMultiNewCombine(_ReprojBase):

    def reproj_loss(self, curr_imgs, prev_imgs, next_imgs, ray_indices, weights, ts, metas, ms_rays, deltas=None):
        bs, num_cams = curr_imgs.shape[:2]
        num_rays = ms_rays.shape[0]
        assert bs == 1
        T_prev = self._transforms(metas, 'img2prevImg', ts[0], num_cams)[0]
        T_next = self._transforms(metas, 'img2nextImg', ts[0], num_cams)[0]
        pix = ms_rays.float().contiguous()
        # The reference loops over the cameras in python (reproj_loss_mono_multi_new_combine.py:108-201).  Only
        # the fused sampling kernel is per camera here; the lattice sampling, SSIM, masks and the minimum run
        # batched over the cameras (same per-element arithmetic, ~6x fewer launches of tiny kernels).
        rgb_curr = self._sample_lattice_all(pix, curr_imgs[0].float(), 'border')                   # (N, R, 3)
        # ray-sharded head: weights / ts hold this rank's rows only — the per-sample kernel runs on them, its three
        # per-ray results are gathered into the full lattice (everything below needs whole images: SSIM, minimum)
        from ..dist import shard_of
        shard = shard_of(weights)
        rays_k = num_rays if shard is None else shard.rays_per_cam_local
        pix_k = pix if shard is None else shard.pix_local.float().contiguous()
        l1s, combs, valids = [], [], []
        for cam, (weight, t) in enumerate(zip(weights, ts)):
            l1, comb, any_valid = ReprojSampleFunction.apply(
                weight.reshape(rays_k, -1), t.reshape(rays_k, -1),
                None if deltas is None else deltas[cam].detach().reshape(rays_k, -1), pix_k,
                rgb_curr[cam] if shard is None else self._local_rows(shard, rgb_curr[cam]).contiguous(),
                T_prev[cam], T_next[cam], prev_imgs[0, cam].float(), next_imgs[0, cam].float(),
                self.img_size[0], self.img_size[1])
            l1s.append(l1); combs.append(comb); valids.append(any_valid)
        l1, comb, any_valid = torch.stack(l1s), torch.stack(combs), torch.stack(valids)             # (N, R[, 3])
        if shard is not None:
            l1, comb, any_valid = self._gather_per_ray(shard, l1, comb, any_valid)
        im = lambda x: x.reshape(num_cams, *self.ray_resize, self.dims).permute(0, 3, 1, 2)
        prev_next = l1
        if not self.no_ssim:
            prev_next = 0.15 * l1 + 0.85 * self.ssim(im(comb), im(rgb_curr)).mean(1).flatten(1)
        if not self.no_automask:
            target_prev = self._sample_lattice_all(pix, prev_imgs[0].float(), 'border')
            target_next = self._sample_lattice_all(pix, next_imgs[0].float(), 'border')
            prev_next = torch.where(any_valid > 0, prev_next, prev_next.new_full((), 1e3))
            proj = torch.stack([prev_next, self._photometric_all(target_prev, rgb_curr, im),
                                self._photometric_all(target_next, rgb_curr, im)], dim=-1).min(dim=-1)[0]
        else:
            proj = prev_next
        tot = proj.mean(dim=1).sum()
        self.iter_counter += 1
        return tot / num_cams


@OPENOCC_LOSS.register_module()
class ReprojLossMonoMultiNew(_ReprojBase):
    """Mono / KITTI variant: the two temporal frames are masked and renormalised separately and
    enter the minimum as two candidates (reproj_loss_mono_multi_new.py:165-255)."""

    def __init__(self, weight=1.0, input_dict=None, **kwargs):
        super().__init__(weight, input_dict, **kwargs)
        if kwargs.get('sdf_loss', False):
            raise NotImplementedError("sdf_loss=True is not used by any shipped config")

    def reproj_loss(self, curr_imgs, prev_imgs, next_imgs, ray_indices, weights, ts, metas, ms_rays, deltas=None,
                    sample_sdfs=None):
        bs, num_cams = curr_imgs.shape[:2]
        num_rays = ms_rays.shape[0]
        assert bs == 1
        T_prev = self._transforms(metas, 'img2prevImg', ts[0], num_cams)[0]
        T_next = self._transforms(metas, 'img2nextImg', ts[0], num_cams)[0]
        pix = ms_rays.float().contiguous()
        invalid = _invalid_on(pix.device)
        from ..dist import shard_of
        shard = shard_of(weights)        # ray-sharded head: per-sample inputs are this rank's rows (see the Combine loss)
        rays_k = num_rays if shard is None else shard.rays_per_cam_local
        pix_k = pix if shard is None else shard.pix_local.float().contiguous()
        tot = 0.
        for cam, (weight, t) in enumerate(zip(weights, ts)):
            target_curr = self._sample_lattice(pix, curr_imgs[0, cam].float(), 'zeros')
            target_k = target_curr if shard is None else self._local_rows(shard, target_curr).contiguous()
            w2, t2 = weight.reshape(rays_k, -1), t.reshape(rays_k, -1)
            d2 = None if deltas is None else deltas[cam].detach().reshape(rays_k, -1)
            cands = []
            for T, img in ((T_prev[cam], prev_imgs[0, cam].float()), (T_next[cam], next_imgs[0, cam].float())):
                l1, comb, any_valid = ReprojSampleFunction.apply(w2, t2, d2, pix_k, target_k, T, invalid, img, img,
                                                                 self.img_size[0], self.img_size[1])
                if shard is not None:      # one camera's rows: gather as a 1-camera lattice
                    from ..render import RaySet
                    one = RaySet(img2lidar=shard.full.img2lidar[:1], nx=shard.full.nx, ny=shard.full.ny)
                    l1, comb, any_valid = (x[0] for x in self._gather_per_ray(
                        type(shard)(one, shard.local, shard.pix_local), l1[None], comb[None], any_valid[None]))
                loss = l1
                if not self.no_ssim:
                    im = lambda x: x.reshape(1, *self.ray_resize, self.dims).permute(0, 3, 1, 2)
                    loss = 0.85 * self.ssim(im(comb), im(target_curr)).mean(1).flatten() + 0.15 * l1
                cands.append(torch.where(any_valid > 0, loss, loss.new_full((), 1e3)))
            if not self.no_automask:
                for img in (prev_imgs[0, cam].float(), next_imgs[0, cam].float()):
                    cands.append(self._photometric(self._sample_lattice(pix, img, 'zeros'), target_curr))
            tot = tot + torch.stack(cands, dim=-1).min(dim=-1)[0].mean()
        self.iter_counter += 1
        return tot / num_cams
