"""Host side of the fused reprojection sampling (selfocc_reproj_fwd / _bwd): the
per-sample part of ReprojLossMonoMultiNewCombine.reproj_loss
(loss/reproj_loss_mono_multi_new_combine.py:108-201) for one camera."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import abi
from ._lib import lib, check, ptr, current_stream


def _args(weights, ts, deltas, pix, curr_rgb, T_prev, T_next, img_prev, img_next, img_h, img_w):
    R, S = weights.shape
    a = abi.SoReprojArgs()
    a.weights, a.ts = ptr(weights), ptr(ts)
    a.deltas = ptr(deltas)
    a.pix, a.curr_rgb = ptr(pix), ptr(curr_rgb)
    a.T_prev, a.T_next = ptr(T_prev), ptr(T_next)
    a.img_prev, a.img_next = ptr(img_prev), ptr(img_next)
    a.R, a.S = R, S
    a.Hi, a.Wi = img_prev.shape[-2], img_prev.shape[-1]
    a.img_h, a.img_w = float(img_h), float(img_w)
    return a


class ReprojSampleFunction(Function):
    """(weights (R,S), ts (R,S), deltas (R,S)|None, pix (R,2), curr_rgb (R,3), T_prev (4,4),
    T_next (4,4), img_prev (3,Hi,Wi), img_next (3,Hi,Wi), img_h, img_w)
       -> l1 (R), rgb_combine (R,3), any_valid (R).   Differentiable wrt ``weights`` only
    (ts / pixels / matrices / images are constants of the loss in the reference too)."""

    @staticmethod
    def forward(ctx, weights, ts, deltas, pix, curr_rgb, T_prev, T_next, img_prev, img_next, img_h, img_w):
        if not weights.is_cuda:
            raise RuntimeError("ReprojSampleFunction needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
        f = lambda t: None if t is None else t.detach().contiguous().float()
        tens = [f(t) for t in (weights, ts, deltas, pix, curr_rgb, T_prev, T_next, img_prev, img_next)]
        a = _args(*tens, img_h, img_w)
        R = tens[0].shape[0]
        dev = weights.device
        l1 = torch.empty(R, device=dev)
        comb = torch.empty(R, 3, device=dev)
        anyv = torch.empty(R, device=dev)
        a.l1, a.rgb_combine, a.any_valid = ptr(l1), ptr(comb), ptr(anyv)
        check(lib().selfocc_reproj_fwd(a, current_stream(dev)), "selfocc_reproj_fwd")
        # saved through autograd (version-counter checks, saved-tensor hooks): an in-place change of `weights`
        # between forward and backward is an error, not a silent mismatch
        ctx.has_deltas = tens[2] is not None
        ctx.save_for_backward(*[t for t in tens if t is not None])
        ctx.hw = (img_h, img_w)
        ctx.mark_non_differentiable(anyv)
        return l1, comb, anyv

    @staticmethod
    @once_differentiable
    def backward(ctx, g_l1, g_comb, _g_any):
        tens = list(ctx.saved_tensors)
        if not ctx.has_deltas:
            tens.insert(2, None)
        a = _args(*tens, *ctx.hw)
        g_l1 = g_l1.contiguous().float()
        g_comb = g_comb.contiguous().float()
        g_w = torch.zeros_like(tens[0])
        check(lib().selfocc_reproj_bwd(a, ptr(g_l1), ptr(g_comb), ptr(g_w), current_stream(g_w.device)),
              "selfocc_reproj_bwd")
        return (g_w,) + (None,) * 10
