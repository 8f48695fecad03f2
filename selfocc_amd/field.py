"""Tri-plane -> dense SDF / feature volume in one HIP kernel (``selfocc_field_volume_fwd``).

Mirror of the head's ``pre_compute_density_color`` for the tri-plane representation
(sdfstudio-fork SDFCustomField, driven from model/head/neus_head/neus_head.py:295-306; in-repo
analogue BEVNeRF, model/head/nerfacc_head/bev_nerf.py:74-95).  Forward only: training keeps the
autograd path in ``SDFField.pre_compute_density_color``."""
import torch

from . import abi
from ._lib import lib, check, ptr, current_stream

SUPPORTED_EMBED_DIMS = (64, 96, 128)


def field_volume_supported(embed_dims, n_linear, out_dim, feat_stride):
    return embed_dims in SUPPORTED_EMBED_DIMS and n_linear in (1, 2) and 1 <= out_dim <= 32 and feat_stride <= 31


def field_volume(hw, zh, wz, size_hwd, linears, feat_stride=0, feat_dtype=torch.float32):
    """hw (H*W, C), zh (D*H, C), wz (W*D, C) float32 CUDA tensors (any leading shape that flattens to
    these); ``linears``: the 1 or 2 ``nn.Linear`` modules of the [Softplus, Linear] x n stack.
    Returns (sdf (H, W, D) float32, feat (H, W, D, feat_stride) or None)."""
    H, W, D = size_hwd
    if not hw.is_cuda:
        raise RuntimeError("field_volume needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
    C = hw.shape[-1]
    hw = hw.reshape(H * W, C).contiguous().float()
    zh = zh.reshape(D * H, C).contiguous().float()
    wz = wz.reshape(W * D, C).contiguous().float()
    out_lin = linears[-1]
    hid = linears[0] if len(linears) == 2 else None
    out_dim = out_lin.weight.shape[0]
    if not field_volume_supported(C, len(linears), out_dim, feat_stride):
        raise ValueError(f"field_volume: unsupported configuration C={C}, {len(linears)} linear layers, out_dim={out_dim}")
    sdf = torch.empty(H, W, D, device=hw.device, dtype=torch.float32)
    feat = torch.empty(H, W, D, feat_stride, device=hw.device, dtype=feat_dtype) if feat_stride > 0 else None
    f32 = lambda t: t.detach().contiguous().float()
    wh, bh = (f32(hid.weight), f32(hid.bias)) if hid is not None else (None, None)
    wo, bo = f32(out_lin.weight), f32(out_lin.bias)
    check(lib().selfocc_field_volume_fwd(ptr(hw), ptr(zh), ptr(wz), H, W, D, C, ptr(wh), ptr(bh),
                                         0 if hid is None else 1, ptr(wo), ptr(bo), out_dim, ptr(sdf), ptr(feat),
                                         abi.DTYPE_BF16 if feat_dtype == torch.bfloat16 else abi.DTYPE_F32,
                                         feat_stride, current_stream(hw.device)),
          "selfocc_field_volume_fwd")
    return sdf, feat


def field_volume_train_supported(embed_dims, n_linear, out_dim, feat_stride, feat_dtype, size_hwd=None):
    ok = embed_dims == 96 and n_linear == 2 and 1 <= out_dim <= 32 and feat_stride <= 31 and feat_dtype == torch.float32
    return ok          # any (H, W, D): the backward walks 4 x 4 x 2 voxel patches (ragged ones are masked)


class FieldVolumeFunction(torch.autograd.Function):
    """Training form: (sdf, feat) = volume MLP of the tri-plane, fused in both directions
    (``selfocc_field_volume_fwd`` / ``selfocc_field_volume_bwd``).  The forward keeps only its inputs — the
    backward kernel recomputes the activations tile by tile — so the (H*W*D, 96) intermediates of the
    op-by-op path (634 MB each at the nuscenes_occ size) never exist."""

    @staticmethod
    def forward(ctx, hw, zh, wz, w1, b1, w2, b2, size_hwd, feat_stride):
        H, W, D = size_hwd
        C = hw.shape[-1]
        ctx.in_shapes = (hw.shape, zh.shape, wz.shape)
        hw, zh, wz = (t.reshape(-1, C).contiguous().float() for t in (hw, zh, wz))
        w1, b1, w2, b2 = (t.contiguous().float() for t in (w1, b1, w2, b2))
        out_dim = w2.shape[0]
        sdf = torch.empty(H, W, D, device=hw.device, dtype=torch.float32)
        feat = torch.empty(H, W, D, feat_stride, device=hw.device, dtype=torch.float32) if feat_stride > 0 else None
        check(lib().selfocc_field_volume_fwd(ptr(hw), ptr(zh), ptr(wz), H, W, D, C, ptr(w1), ptr(b1), 1, ptr(w2), ptr(b2),
                                             out_dim, ptr(sdf), ptr(feat), abi.DTYPE_F32, feat_stride,
                                             current_stream(hw.device)), "selfocc_field_volume_fwd")
        ctx.save_for_backward(hw, zh, wz, w1, b1, w2)
        ctx.dims = (H, W, D, C, out_dim, feat_stride)
        if feat is None:
            return sdf, sdf.new_zeros(0)
        return sdf, feat

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_feat):
        hw, zh, wz, w1, b1, w2 = ctx.saved_tensors
        H, W, D, C, out_dim, F = ctx.dims
        g_sdf = None if g_sdf is None else g_sdf.contiguous().float()
        g_feat = None if (g_feat is None or F == 0) else g_feat.contiguous().float()
        g_hw, g_zh, g_wz = torch.zeros_like(hw), torch.zeros_like(zh), torch.zeros_like(wz)
        g_w1, g_b1, g_w2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2)
        g_b2 = torch.zeros(out_dim, device=hw.device, dtype=torch.float32)
        check(lib().selfocc_field_volume_bwd(ptr(hw), ptr(zh), ptr(wz), H, W, D, C, ptr(w1), ptr(b1), ptr(w2), out_dim,
                                             ptr(g_sdf), ptr(g_feat), F, ptr(g_hw), ptr(g_zh), ptr(g_wz), ptr(g_w1),
                                             ptr(g_b1), ptr(g_w2), ptr(g_b2), current_stream(hw.device)),
              "selfocc_field_volume_bwd")
        s_hw, s_zh, s_wz = ctx.in_shapes
        return g_hw.view(s_hw), g_zh.view(s_zh), g_wz.view(s_wz), g_w1, g_b1, g_w2, g_b2, None, None
