"""Multi-scale deformable attention op — same name, signature and autograd contract as
``mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction`` (mmcv==2.0.1), the
native op the reference calls at
model/encoder/bevformer/attention/image_cross_attention.py:340-342 and
model/encoder/tpvformer/attention/cross_view_hybrid_attention.py:111-113.
The arithmetic is csrc/msda.hip behind selfocc_msda_fwd / selfocc_msda_bwd.
"""
import ctypes as C
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import abi
from ._lib import lib, check, ptr, current_stream


def _prep(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    if not value.is_cuda:
        raise RuntimeError("MultiScaleDeformableAttnFunction needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
    bs, nv, heads, d = value.shape
    _, nq, heads2, L, P, two = sampling_locations.shape
    assert heads2 == heads and two == 2
    assert tuple(attention_weights.shape) == (bs, nq, heads, L, P)
    assert spatial_shapes.shape == (L, 2) and level_start_index.shape == (L,)
    value = value.contiguous().float()
    # mmcv casts locations / weights to value's dtype (amp switch = dtype of value)
    loc = sampling_locations.contiguous().float()
    aw = attention_weights.contiguous().float()
    sh, st = _i32(spatial_shapes, value.device), _i32(level_start_index, value.device)
    return value, sh, st, loc, aw, (bs, nv, nq, heads, d, L, P)


# 'banded': selfocc_msda_bwd_banded (LDS f64 accumulation, default); 'atomic': selfocc_msda_bwd (global float
# atomics; what a C caller without host shapes / workspace gets).  Same gradients up to summation order.
BACKWARD_MODE = 'banded'


class MultiScaleDeformableAttnFunction(Function):

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step=64):
        """value (bs, num_keys, num_heads, embed_dims // num_heads); value_spatial_shapes
        (num_levels, 2) [H, W]; sampling_locations (bs, num_queries, num_heads, num_levels,
        num_points, 2) in [0, 1] (x, y); attention_weights (bs, num_queries, num_heads,
        num_levels, num_points).  Returns (bs, num_queries, embed_dims).  ``im2col_step`` is
        accepted for signature compatibility; the HIP kernel needs no batch chunking."""
        value, sh, st, loc, aw, dims = _prep(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights)
        bs, nv, nq, heads, d, L, P = dims
        out = torch.empty(bs, nq, heads * d, device=value.device, dtype=torch.float32)
        check(lib().selfocc_msda_fwd(ptr(value), ptr(sh), ptr(st), ptr(loc), ptr(aw), ptr(out),
                                     bs, nv, nq, heads, d, L, P, current_stream(value.device)),
              "selfocc_msda_fwd")
        ctx.save_for_backward(value, sh, st, loc, aw)
        ctx.dims = dims
        # host copy of the level shapes for the banded backward's work decomposition (no device read-back
        # when the caller hands over CPU / python shapes; one small sync otherwise, in forward only)
        host = getattr(value_spatial_shapes, '_so_host', None)
        if host is None and ctx.needs_input_grad[0]:
            host = [int(v) for v in value_spatial_shapes.reshape(-1).tolist()]
        ctx.host_shapes = host if ctx.needs_input_grad[0] else None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, sh, st, loc, aw = ctx.saved_tensors
        bs, nv, nq, heads, d, L, P = ctx.dims
        g_out = grad_output.contiguous().float()
        g_value = torch.zeros_like(value)
        g_loc = torch.empty_like(loc)
        g_aw = torch.empty_like(aw)
        if ctx.host_shapes is not None and L <= 8 and BACKWARD_MODE == 'banded':
            import ctypes
            arr = (ctypes.c_int32 * len(ctx.host_shapes))(*ctx.host_shapes)
            nbytes = int(lib().selfocc_msda_bwd_banded_workspace(bs, nq, heads, L, P))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=value.device)
            check(lib().selfocc_msda_bwd_banded(ptr(value), ptr(sh), ptr(st), ctypes.cast(arr, ctypes.c_void_p),
                                                ptr(loc), ptr(aw), ptr(g_out), ptr(g_value), ptr(g_loc), ptr(g_aw),
                                                bs, nv, nq, heads, d, L, P, ptr(ws), nbytes,
                                                current_stream(value.device)),
                  "selfocc_msda_bwd_banded")
        else:
            check(lib().selfocc_msda_bwd(ptr(value), ptr(sh), ptr(st), ptr(loc), ptr(aw), ptr(g_out),
                                         ptr(g_value), ptr(g_loc), ptr(g_aw),
                                         bs, nv, nq, heads, d, L, P, current_stream(value.device)),
                  "selfocc_msda_bwd")
        return g_value, None, None, g_loc, g_aw, None


def multi_scale_deformable_attn(value, spatial_shapes, level_start_index, sampling_locations,
                                attention_weights):
    return MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index,
                                                  sampling_locations, attention_weights, 64)


def _i32(t, device):
    """int32 device copy of a small index tensor (level shapes / starts), cached on the tensor object: the same
    constants go through 16 MSDA calls per frame and each conversion is a kernel launch."""
    if t.dtype == torch.int32 and t.device == device and t.is_contiguous():
        return t
    if t.is_inference():
        # inference tensors carry no version counter (reading ``_version`` raises): they cannot be written in place
        # outside inference mode either, so the object identity is the cache key
        ver = -1
    else:
        ver = t._version
    c = getattr(t, '_so_i32', None)
    if c is None or c[0] != ver or c[1].device != device:
        c = (ver, t.to(device=device, dtype=torch.int32).contiguous())
        try:
            t._so_i32 = c
        except AttributeError:
            pass
    return c[1]


def _u8(mask):
    """bool / uint8 mask as contiguous uint8 without a conversion kernel for bool (same storage)."""
    mask = mask.contiguous()
    return mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)


def to_head_major(value):
    """(bs, nv, h, d) -> a dense (bs, h, nv, d) copy: the layout the fused / camera-loop kernels gather fastest from
    (a cache line holds x-neighbours of one head instead of one pixel of two heads; include/selfocc_hip.h)."""
    return value.permute(0, 2, 1, 3).contiguous()


def _value_arg(value):
    """(contiguous tensor, SO_DTYPE): bfloat16 storage is passed through, everything else as float32."""
    if value.dtype == torch.bfloat16:
        return value.contiguous(), abi.DTYPE_BF16
    return value.contiguous().float(), abi.DTYPE_F32


def _value_dims(value, head_major):
    if head_major:
        bs, heads, nv, d = value.shape
    else:
        bs, nv, heads, d = value.shape
    return bs, nv, heads, d


def _off_logits_args(sampling_offsets, attention_logits, heads, merged_LP):
    """(off tensor kept alive, off pointer, logits pointer, nq-leading shape, L, P, ol_stride) for the fused / camera-loop
    entry points.  Two forms: the dense pair ``sampling_offsets (..., nq, h, L, P, 2)`` + ``attention_logits (..., nq, h, L*P)``,
    or — ``attention_logits is None`` — ONE merged projection output ``sampling_offsets (..., nq, 3 * h * L * P)`` whose rows
    are [h*L*P*2 raw offsets | h*L*P logits] (the stacked sampling_offsets | attention_weights Linear, ABI 32 ``ol_stride``);
    ``merged_LP = (L, P)``."""
    import ctypes
    if attention_logits is not None:
        off = sampling_offsets.contiguous().float()
        lg = attention_logits.contiguous().float()
        L, P = off.shape[-3], off.shape[-2]
        return (off, lg), ptr(off), ptr(lg), off.shape[:-4], L, P, 0
    L, P = merged_LP
    ol = sampling_offsets
    n = 3 * heads * L * P
    assert ol.shape[-1] == n and ol.dtype == torch.float32, (tuple(ol.shape), n)
    if ol.stride(-1) != 1 or (ol.dim() > 1 and ol.stride(-2) % 2) or not ol.reshape(-1, n).is_contiguous():
        ol = ol.contiguous()
    return (ol,), ptr(ol), ctypes.c_void_p(ol.data_ptr() + 8 * heads * L * P), ol.shape[:-1], L, P, n


def msda_fused_inference(value, spatial_shapes, level_start_index, reference_points, ref_kind, sampling_offsets,
                         attention_logits, head_major=False, merged_LP=None):
    """Inference-only fused op (no autograd): value (bs,nv,h,d), or (bs,h,nv,d) with ``head_major``;
    reference_points per ``ref_kind``
    (0: (bs,nq,L,2), 1: (bs,nq,P,2), 2: (bs,nq,L,P,2)); sampling_offsets (bs,nq,h,L,P,2) raw linear
    output; attention_logits (bs,nq,h,L*P) before softmax.  Returns (bs, nq, h*d)."""
    if not value.is_cuda:
        raise RuntimeError("msda_fused_inference needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
    bs, nv, heads, d = _value_dims(value, head_major)
    keep, p_off, p_lg, lead, L, P, ols = _off_logits_args(sampling_offsets, attention_logits, heads, merged_LP)
    nq = lead[-1]
    value, vdt = _value_arg(value)
    ref = reference_points.contiguous().float()
    sh, st = _i32(spatial_shapes, value.device), _i32(level_start_index, value.device)
    out = torch.empty(bs, nq, heads * d, device=value.device, dtype=torch.float32)
    check(lib().selfocc_msda_fused_fwd(ptr(value), ptr(sh), ptr(st), ptr(ref), int(ref_kind), p_off, p_lg,
                                       ptr(out), bs, nv, nq, heads, d, L, P, int(bool(head_major)), vdt, ols,
                                       current_stream(value.device)),
          "selfocc_msda_fused_fwd")
    return out


def msda_cross_inference(value, spatial_shapes, level_start_index, reference_points_cam, visible,
                         sampling_offsets, attention_logits, head_major=False, merged_LP=None):
    """Inference-only camera-loop op (no autograd): the sampling stage of BEVCrossAttention
    (bevformer/attention/image_cross_attention.py:90-136) without re-batching.
    value (cams,nv,h,d), or (cams,h,nv,d) with ``head_major``; reference_points_cam (cams,nq,P,2); visible (cams,nq) bool — the cameras that see
    each query; sampling_offsets (nq,h,L,P,2) and attention_logits (nq,h,L*P): the raw linear outputs for
    the UN-rebatched queries.  Returns (nq, h*d): the mean over the visible cameras."""
    if not value.is_cuda:
        raise RuntimeError("msda_cross_inference needs CUDA(HIP) tensors: selfocc_amd has no CPU fallback")
    cams, nv, heads, d = _value_dims(value, head_major)
    keep, p_off, p_lg, lead, L, P, ols = _off_logits_args(sampling_offsets, attention_logits, heads, merged_LP)
    nq = lead[-1]
    vstride = 0
    vdt = abi.DTYPE_BF16 if value.dtype == torch.bfloat16 else abi.DTYPE_F32
    if (not head_major and value.dtype in (torch.float32, torch.bfloat16) and not value.is_contiguous()
            and value.stride(3) == 1 and value.stride(2) == d
            and value.stride(1) % 4 == 0 and value.stride(0) == nv * value.stride(1)):
        vstride = value.stride(1)        # a column block of a wider (cams * nv, N) matrix: no copy
    else:
        value, vdt = _value_arg(value)
    ref = reference_points_cam.contiguous().float()
    vis = _u8(visible)
    assert ref.shape == (cams, nq, P, 2) and vis.shape == (cams, nq)
    sh, st = _i32(spatial_shapes, value.device), _i32(level_start_index, value.device)
    out = torch.empty(nq, heads * d, device=value.device, dtype=torch.float32)
    check(lib().selfocc_msda_cross_fwd(ptr(value), ptr(sh), ptr(st), ptr(ref), ptr(vis), p_off, p_lg,
                                       ptr(out), cams, nv, nq, heads, d, L, P, vstride, int(bool(head_major)), vdt, ols,
                                       current_stream(value.device)),
          "selfocc_msda_cross_fwd")
    return out


def _grad_value_target(grad_sink, value, B, nv, heads, d):
    """(tensor returned as grad_value, pointer handed to the kernel, g_value_stride): a fresh zeroed tensor in the layout of
    ``value``, or — with a ValueGradSink — this attention's column block of the shared row-major buffer"""
    import ctypes
    if grad_sink is None:
        g_value = torch.zeros(value.shape, device=value.device, dtype=torch.float32)
        return g_value, ptr(g_value), 0
    sink, g = grad_sink
    buf = sink.slot(g, B, nv, heads, d, value.device)
    return (buf[:, :, g].permute(0, 2, 1, 3), ctypes.c_void_p(buf.data_ptr() + 4 * g * heads * d), sink.G * heads * d)


class ValueGradSink:
    """Where the MSDA backward of G attentions that share ONE stacked value projection put their grad_value: a single
    row-major buffer (B, nv, G, heads, d) — the dy of that projection's weight- / input-gradient passes — written PIXEL-major by
    the band kernels themselves (ABI 32 ``g_value_stride``).  Until round 6 every attention returned a head-major grad_value
    and the projection's backward re-assembled the rows with one transposing copy per attention (12 x 59 MB copies per
    nuscenes_occ iteration).  Created by bricks.value_proj_head_major[_multi], handed to the attention's MSDA Function as
    ``grad_sink=(sink, g)``; the Function returns a (B, heads, nv, d) VIEW of the buffer as the gradient of its head-major
    input, and the projection's backward recognises the views (``rows()``) and uses the buffer as it is."""

    hits = 0        # backward passes that used the shared buffer as it was (tests read this)

    def __init__(self, G):
        self.G, self.buf, self.filled = G, None, set()

    def slot(self, g, B, nv, heads, d, device):
        if self.buf is None or g in self.filled:       # a new backward pass (or the same attention twice): a fresh buffer
            self.buf, self.filled = torch.zeros(B, nv, self.G, heads, d, device=device, dtype=torch.float32), set()
        self.filled.add(g)
        return self.buf

    def rows(self, grads):
        """the (B * nv, G * heads * d) matrix when ``grads`` are exactly this buffer's G views, else None; releases the buffer"""
        buf, self.buf, self.filled = self.buf, None, set()
        if buf is None or len(grads) != self.G:
            return None
        for g, t in enumerate(grads):
            v = buf[:, :, g].permute(0, 2, 1, 3)
            if t is None or t.data_ptr() != v.data_ptr() or t.shape != v.shape or any(
                    a != b for a, b, n in zip(t.stride(), v.stride(), v.shape) if n > 1):      # (size-1 dims: any stride)
                return None
        ValueGradSink.hits += 1
        return buf.view(buf.shape[0] * buf.shape[1], -1)


class MSDAFusedFunction(torch.autograd.Function):
    """Training form of the fused op: out = MSDA(value, ref + off / (W_l, H_l), softmax(logits)) with the
    prologue inside the kernels in BOTH directions.  The forward saves only its inputs (no sampling_locations /
    attention_weights tensors); the backward returns gradients w.r.t. value, the raw offsets and the raw
    logits (``selfocc_msda_fused_bwd``).  Same math as softmax -> loc -> MultiScaleDeformableAttnFunction."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, reference_points, ref_kind, sampling_offsets,
                attention_logits, host_shapes, head_major=False, value_bf16=False, merged_LP=None, grad_sink=None):
        """``attention_logits is None``: ``sampling_offsets`` is the merged projection output (bs, nq, 3 h L P) (rows
        [offsets | logits], ``merged_LP = (L, P)``) and the backward returns ONE gradient of that shape.
        ``grad_sink = (ValueGradSink, g)`` (head-major float32 value only): grad_value is written pixel-major into the sink."""
        ctx.grad_sink = grad_sink if (head_major and not value_bf16) else None
        if value_bf16:      # bfloat16 STORAGE of value for the gathers (forward and backward); gradients stay float32
            value = value.to(torch.bfloat16)
        out = msda_fused_inference(value, spatial_shapes, level_start_index, reference_points, ref_kind,
                                   sampling_offsets, attention_logits, head_major, merged_LP)
        ctx.head_major = bool(head_major)
        sh, st = _i32(spatial_shapes, value.device), _i32(level_start_index, value.device)
        ctx.merged_LP = merged_LP if attention_logits is None else None
        if ctx.merged_LP is None:
            ctx.save_for_backward(value, sh, st, reference_points, sampling_offsets, attention_logits)
        else:
            ctx.save_for_backward(value, sh, st, reference_points, sampling_offsets)
        ctx.ref_kind, ctx.host_shapes = int(ref_kind), list(host_shapes)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        import ctypes
        if ctx.merged_LP is None:
            value, sh, st, ref, off, lg = ctx.saved_tensors
        else:
            (value, sh, st, ref, off), lg = ctx.saved_tensors, None
        ref = ref.contiguous().float()
        value, vdt = _value_arg(value)
        bs, nv, heads, d = _value_dims(value, ctx.head_major)
        keep, p_off, p_lg, lead, L, P, ols = _off_logits_args(off, lg, heads, ctx.merged_LP)
        nq = lead[-1]
        g_out = grad_output.contiguous().float()
        g_value, p_gv, gvs = _grad_value_target(ctx.grad_sink, value, bs, nv, heads, d)
        if ols:
            g_off = torch.empty(*lead, ols, device=value.device, dtype=torch.float32)
            g_lg, pg_lg = None, ctypes.c_void_p(g_off.data_ptr() + 8 * heads * L * P)
        else:
            g_off, g_lg = torch.empty_like(keep[0]), torch.empty_like(keep[1])
            pg_lg = ptr(g_lg)
        arr = (ctypes.c_int32 * len(ctx.host_shapes))(*ctx.host_shapes)
        nbytes = int(lib().selfocc_msda_bwd_banded_workspace(bs, nq, heads, L, P))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=value.device)
        check(lib().selfocc_msda_fused_bwd(ptr(value), ptr(sh), ptr(st), ctypes.cast(arr, ctypes.c_void_p), ptr(ref),
                                           ctx.ref_kind, p_off, p_lg, ptr(g_out), p_gv, ptr(g_off),
                                           pg_lg, bs, nv, nq, heads, d, L, P, int(ctx.head_major), vdt, ols, gvs, ptr(ws), nbytes,
                                           current_stream(value.device)),
              "selfocc_msda_fused_bwd")
        return g_value, None, None, None, None, g_off, g_lg, None, None, None, None, None


def msda_fused_kernels_built(d, value_bf16=False):
    """The fused / camera-loop kernels exist for 8, 16, 32 channels per head; a bfloat16 ``value`` for 16 only
    (csrc/msda.hip: SO_FUSED_DISPATCH).  Everything else goes through the plain op (4, 8, 16, 32; float32)."""
    return d in (8, 16, 32) and (not value_bf16 or d == 16)


def msda_fused_supported(host_shapes, bs, nq, heads, d, L, P, value_bf16=False):
    """True when the fused training op applies (banded scatter possible, L * P <= 256)."""
    import ctypes
    if L * P > 256 or L > 8 or not msda_fused_kernels_built(d, value_bf16):
        return False
    arr = (ctypes.c_int32 * len(host_shapes))(*host_shapes)
    return lib().selfocc_msda_banded_supported(ctypes.cast(arr, ctypes.c_void_p), bs, nq, heads, d, L, P) == 1


class MSDACrossFunction(torch.autograd.Function):
    """Training form of ``msda_cross_inference``: the camera-loop sampling stage of BEVCrossAttention under
    autograd.  Differentiable inputs: value (cams,nv,h,d), sampling_offsets (nq,h,L,P,2), attention_logits
    (nq,h,L*P); the reference points and the visibility mask are geometry (no gradient)."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, reference_points_cam, visible, sampling_offsets,
                attention_logits, host_shapes, head_major=False, value_bf16=False, merged_LP=None, grad_sink=None):
        """``attention_logits is None``: ``sampling_offsets`` is the merged projection output (nq, 3 h L P); ``grad_sink``: see
        MSDAFusedFunction."""
        ctx.grad_sink = grad_sink if (head_major and not value_bf16) else None
        if value_bf16:
            value = value.to(torch.bfloat16)
        out = msda_cross_inference(value, spatial_shapes, level_start_index, reference_points_cam, visible,
                                   sampling_offsets, attention_logits, head_major, merged_LP)
        ctx.head_major = bool(head_major)
        sh, st = _i32(spatial_shapes, value.device), _i32(level_start_index, value.device)
        ctx.merged_LP = merged_LP if attention_logits is None else None
        if ctx.merged_LP is None:
            ctx.save_for_backward(value, sh, st, reference_points_cam, _u8(visible), sampling_offsets, attention_logits)
        else:
            ctx.save_for_backward(value, sh, st, reference_points_cam, _u8(visible), sampling_offsets)
        ctx.host_shapes = list(host_shapes)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        import ctypes
        if ctx.merged_LP is None:
            value, sh, st, ref, vis, off, lg = ctx.saved_tensors
        else:
            (value, sh, st, ref, vis, off), lg = ctx.saved_tensors, None
        ref = ref.contiguous().float()
        value, vdt = _value_arg(value)
        vis = vis.contiguous()
        cams, nv, heads, d = _value_dims(value, ctx.head_major)
        keep, p_off, p_lg, lead, L, P, ols = _off_logits_args(off, lg, heads, ctx.merged_LP)
        nq = lead[-1]
        g_out = grad_output.contiguous().float()
        g_value, p_gv, gvs = _grad_value_target(ctx.grad_sink, value, cams, nv, heads, d)
        if ols:
            g_off = torch.empty(*lead, ols, device=value.device, dtype=torch.float32)
            g_lg, pg_lg = None, ctypes.c_void_p(g_off.data_ptr() + 8 * heads * L * P)
        else:
            g_off, g_lg = torch.empty_like(keep[0]), torch.empty_like(keep[1])
            pg_lg = ptr(g_lg)
        arr = (ctypes.c_int32 * len(ctx.host_shapes))(*ctx.host_shapes)
        nbytes = int(lib().selfocc_msda_bwd_banded_workspace(cams, nq, heads, L, P))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=value.device)
        check(lib().selfocc_msda_cross_bwd(ptr(value), ptr(sh), ptr(st), ctypes.cast(arr, ctypes.c_void_p), ptr(ref),
                                           ptr(vis), p_off, p_lg, ptr(g_out), p_gv, ptr(g_off),
                                           pg_lg, cams, nv, nq, heads, d, L, P, int(ctx.head_major), vdt, ols, gvs, ptr(ws),
                                           nbytes, current_stream(value.device)),
              "selfocc_msda_cross_bwd")
        return g_value, None, None, None, None, g_off, g_lg, None, None, None, None, None
