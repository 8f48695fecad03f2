"""Small building blocks the reference takes from mmengine / mmcv (absent here): BaseModule,
ModuleList, init helpers, FFN, norm builder and the mmcv MultiScaleDeformableAttention base
class whose parameters / state-dict keys CrossViewHybridAttention inherits
(model/encoder/tpvformer/attention/cross_view_hybrid_attention.py:12)."""
import math
import os
import warnings

import torch
import torch.nn as nn

from ..registry import MODELS
from ..dropout import dropout_add
from ..linear import (linear_wgrad, wgrad_supported, linear_fwd, linear_fwd_supported, linear_fwd_heads, linear_dgrad, dgrad_supported,
                      linear_fwd_heads_supported)
from ..msda import (MultiScaleDeformableAttnFunction, msda_fused_inference, MSDAFusedFunction, to_head_major,
                    msda_fused_supported, msda_fused_kernels_built, ValueGradSink)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        self._is_init = False

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()
        self._is_init = True


ModuleList = nn.ModuleList


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if getattr(module, 'weight', None) is not None:
        (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


class _LayerNormFunction(torch.autograd.Function):
    """LayerNorm over the channel dimension through selfocc_layernorm_fwd / _bwd (csrc/layernorm.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        from .._lib import lib, check, ptr, current_stream
        C = x.shape[-1]
        x2 = x.reshape(-1, C).contiguous().float()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        mean = torch.empty(rows, device=x.device) if need else None
        rstd = torch.empty(rows, device=x.device) if need else None
        w, b = weight.contiguous().float(), bias.contiguous().float()
        check(lib().selfocc_layernorm_fwd(ptr(x2), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, C, float(eps),
                                          current_stream(x.device)), "selfocc_layernorm_fwd")
        if need:
            ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        from .._lib import lib, check, ptr, current_stream
        x2, w, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        dy2 = dy.reshape(rows, C).contiguous().float()
        dx = torch.empty_like(x2)
        dg, db = torch.empty_like(w), torch.empty_like(w)
        nbytes = int(lib().selfocc_layernorm_bwd_workspace(rows, C))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x2.device)
        check(lib().selfocc_layernorm_bwd(ptr(x2), ptr(w), ptr(mean), ptr(rstd), ptr(dy2), ptr(dx), ptr(dg), ptr(db), rows, C,
                                          ptr(ws), nbytes, current_stream(x2.device)), "selfocc_layernorm_bwd")
        return dx.view(ctx.shape), dg, db, None


class FastLayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state-dict keys) whose CUDA float32 forward / backward run the streaming HIP
    kernels of csrc/layernorm.hip; anything else (CPU tensors in the host-side tests, exotic widths) is nn.LayerNorm."""

    def forward(self, x):
        C = x.shape[-1]
        if (x.is_cuda and x.dtype == torch.float32 and self.elementwise_affine and self.bias is not None
                and len(self.normalized_shape) == 1 and C % 4 == 0 and 4 <= C <= 128 and not torch.is_autocast_enabled()):
            return _LayerNormFunction.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)


def build_norm_layer(cfg, num_features):
    typ = cfg.get('type', 'LN')
    if typ != 'LN':
        raise NotImplementedError(f"norm {typ}: only LN is used on the SelfOcc hot path")
    return 'ln', FastLayerNorm(num_features, eps=cfg.get('eps', 1e-5))


def build_activation_layer(cfg):
    typ = cfg.get('type', 'ReLU')
    if typ == 'ReLU':
        return nn.ReLU(inplace=cfg.get('inplace', False))
    if typ == 'GELU':
        return nn.GELU()
    raise NotImplementedError(typ)


# weight / bias gradients of _TallLinear through selfocc_linear_wgrad (False: batched GEMMs + torch reductions, A/B)
FUSED_WGRAD = True
# input gradient of the tall Linears on the bf16x3 kernel (csrc/linear_fwd.hip, selfocc_linear_dgrad) instead of the vendor GEMM
FUSED_DGRAD = os.environ.get('SELFOCC_LINEAR_DGRAD', '1') == '1'
DGRAD_MIN_ROWS = 32768      # below: too few 32-row tiles for 1 024 SIMDs, the vendor GEMM splits the reduction
# forward of the tall projections (and, in inference, the residual add / LayerNorm that follow them) through
# selfocc_linear_fwd (csrc/linear_fwd.hip); False: torch.addmm + separate elementwise kernels (A/B)
FUSED_LINEAR_FWD = os.environ.get('SELFOCC_FUSED_LINEAR', '1') == '1'
LINEAR_FWD_MIN_ROWS = 1024


# inference: value_proj writes the head-major layout the MSDA kernels gather fastest from (selfocc_linear_fwd_heads: no
# transposing copy, unlike HEAD_MAJOR_VALUE below) when the attention has 6 heads x 16 channels (every shipped config)
HEAD_MAJOR_PROJ = os.environ.get('SELFOCC_HEAD_MAJOR_PROJ', '1') == '1'


HEAD_MAJOR_PROJ_TRAIN = os.environ.get('SELFOCC_HEAD_MAJOR_PROJ_TRAIN', '1') == '1'


def value_proj_head_major(lin_weight, lin_bias, value2d, nv, num_heads):
    """(G, B, 6, nv, 16) head-major projection of (B * nv, K) rows through the stacked weight (G * 96, K), or None when
    the shape / mode does not qualify (the caller then projects pixel-major as the reference does).  Under autograd
    the result carries the projection's backward (_TallLinearHeads)."""
    if not (HEAD_MAJOR_PROJ and FUSED_LINEAR_FWD and not torch.is_autocast_enabled()
            and value2d.is_cuda and value2d.dtype == torch.float32 and lin_weight.dtype == torch.float32 and num_heads == 6):
        return None
    rows, n_out = value2d.shape[0], lin_weight.shape[0]
    if rows < LINEAR_FWD_MIN_ROWS or not linear_fwd_heads_supported(rows, n_out, value2d.shape[1], nv):
        return None
    if torch.is_grad_enabled() and (value2d.requires_grad or lin_weight.requires_grad):
        if not HEAD_MAJOR_PROJ_TRAIN:
            return None
        G = n_out // 96
        sink = ValueGradSink(G) if VALUE_GRAD_SINK else None
        out = _TallLinearHeads.apply(value2d, lin_weight, lin_bias, nv, sink)
        if sink is not None:
            out._so_grad_sink = sink           # read by the caller, which hands (sink, g) to the g-th attention's MSDA Function
        return out
    return linear_fwd_heads(value2d, lin_weight, lin_bias, nv)


# training: the MSDA backward writes grad_value pixel-major straight into the row-major gradient of the (stacked) value
# projection (msda.ValueGradSink, ABI 32 g_value_stride) instead of head-major + one transposing copy per attention
VALUE_GRAD_SINK = os.environ.get('SELFOCC_VALUE_GRAD_SINK', '1') == '1'


# training: the three TPV planes' value projections of the same image features as ONE projection / ONE backward
# (_TallLinearHeadsMulti): the 59 MB input is read once per layer instead of three times in each direction, one input
# gradient instead of three + two accumulations (env SELFOCC_MERGED_VALUE_PROJ_TRAIN=0: per plane, as round 2)
MERGED_VALUE_PROJ_TRAIN = os.environ.get('SELFOCC_MERGED_VALUE_PROJ_TRAIN', '1') == '1'


def value_proj_head_major_multi(lins, value2d, nv, num_heads):
    """[(B, 6, nv, 16)] * G: the head-major projections of (B * nv, K) rows through G nn.Linear(K, 96) in one launch
    under autograd, or None when the shape / mode does not qualify (the caller then projects per plane)."""
    if not (HEAD_MAJOR_PROJ and HEAD_MAJOR_PROJ_TRAIN and MERGED_VALUE_PROJ_TRAIN and FUSED_LINEAR_FWD
            and not torch.is_autocast_enabled() and torch.is_grad_enabled() and value2d.is_cuda
            and value2d.dtype == torch.float32 and num_heads == 6):
        return None
    K = value2d.shape[1]
    if any(l.bias is None or tuple(l.weight.shape) != (96, K) or l.weight.dtype != torch.float32 for l in lins):
        return None
    rows = value2d.shape[0]
    if rows < LINEAR_FWD_MIN_ROWS or not linear_fwd_heads_supported(rows, 96 * len(lins), K, nv):
        return None
    if not (value2d.requires_grad or any(l.weight.requires_grad for l in lins)):
        return None
    wb = [t for l in lins for t in (l.weight, l.bias)]
    sink = ValueGradSink(len(lins)) if VALUE_GRAD_SINK else None
    outs = list(_TallLinearHeadsMulti.apply(value2d, nv, sink, *wb))
    if sink is not None:
        for g, o in enumerate(outs):
            o._so_grad_sink = (sink, g)        # read by BEVCrossAttention._forward_camera_loop
    return outs


def _linear_fwd_ok(x2d, weight):
    return (x2d.is_cuda and x2d.dtype == torch.float32 and weight.dtype == torch.float32
            and x2d.shape[0] >= LINEAR_FWD_MIN_ROWS and linear_fwd_supported(x2d.shape[0], weight.shape[0], x2d.shape[1]))


def fused_linear(lin, x, relu=False, residual=None, norm=None, out=None):
    """Inference form of ``norm(relu(lin(x)) + residual)`` (each part optional) for (..., K) activations: ONE
    selfocc_linear_fwd launch when the shape qualifies (CUDA float32, >= LINEAR_FWD_MIN_ROWS rows, K in 32 / 64 / 96 /
    128 / 192, LayerNorm width <= 96), else the separate torch / HIP steps in the reference's order (mmcv FFN:
    ``identity + layers(x)``; image_cross_attention.py:136 ``dropout(slots) + residual``; then the layer's `norm`).
    ``out``: optional (..., N) destination with unit column stride (a slice of a larger buffer)."""
    N, K = lin.weight.shape
    lead = x.shape[:-1]
    rows = x.numel() // max(K, 1)
    x2 = x.reshape(rows, K)
    ln_ok = norm is None or (isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None
                             and tuple(norm.normalized_shape) == (N,) and N <= 96)
    if (FUSED_LINEAR_FWD and not torch.is_grad_enabled() and ln_ok and _linear_fwd_ok(x2, lin.weight)
            and not torch.is_autocast_enabled()):
        r2 = residual.reshape(rows, N) if residual is not None else None
        o2 = None
        if out is not None and out.stride(-1) == 1:
            o2 = out.view(rows, N) if out.is_contiguous() else (out[0] if out.dim() == 3 and out.shape[0] == 1 else None)
        ln = (norm.weight, norm.bias, norm.eps) if norm is not None else None
        y = linear_fwd(x2, lin.weight, lin.bias, relu=relu, residual=r2, ln=ln, out=o2)
        if o2 is not None:
            return out
        y = y.view(*lead, N)
        if out is not None:
            out.copy_(y)
            return out
        return y
    y = lin(x)
    if relu:
        y = torch.relu(y)
    if residual is not None:
        y = y + residual
    if norm is not None:
        y = norm(y)
    if out is not None:
        out.copy_(y)
        return out
    return y


class _TallLinear(torch.autograd.Function):
    """y = x W^T + b for x with millions of rows and a handful of output features.  The vendor
    GEMM picked for the weight gradient dW = dy^T x of such a shape (25 x 1.65 M x 96 at the shipped
    nuscenes_occ sizes) runs on 2 workgroups — 23 ms per call, 4 calls per iteration in the
    round-1 profile; here the reduction over rows is split into 256 batched GEMMs + a sum."""

    # under torch.autocast (the reference's env amp=true) the op runs in float32 like the HIP ops around it
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if FUSED_LINEAR_FWD and _linear_fwd_ok(x, weight):
            return linear_fwd(x, weight, bias)
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        return _tall_linear_backward(x, weight, dy, ctx.has_bias, ctx.needs_input_grad[0])


def _dgrad(dy, weight):
    if (FUSED_DGRAD and dy.is_cuda and dy.dtype == torch.float32 and dy.shape[0] >= DGRAD_MIN_ROWS
            and dgrad_supported(dy.shape[0], weight.shape[0], weight.shape[1])):
        return linear_dgrad(dy, weight)
    return dy @ weight


def _tall_linear_backward(x, weight, dy, has_bias, need_dx=True):
        """(dx, dW, db) of y = x W^T + b for a row-major dy (shared by _TallLinear and _TallLinearHeads)."""
        dy = dy.contiguous().to(x.dtype)
        T = x.shape[0]
        if (FUSED_WGRAD and dy.is_cuda and dy.dtype == torch.float32 and T > 0
                and wgrad_supported(T, dy.shape[1], x.shape[1])):
            # one MFMA pass over dy and x for dW and db (csrc/linear.hip) instead of batched GEMMs + two reductions
            dw, db = linear_wgrad(dy, x, has_bias)
            return (_dgrad(dy, weight) if need_dx else None), dw, db
        G = max(1, min(256, T // 2048))  # ~2 k+ rows per batched GEMM: enough workgroups, small partial-sum tensor
        R = T // G                       # rows per batched GEMM
        Tp = R * G
        dw = dy.new_zeros(dy.shape[1], x.shape[1])
        db = dy.new_zeros(dy.shape[1])
        if R > 0:
            dyb = dy[:Tp].view(G, R, dy.shape[1])
            dw = torch.bmm(dyb.transpose(1, 2), x[:Tp].view(G, R, x.shape[1]))
            dw = dw[0] if G == 1 else dw.sum(0)
            # bias gradient in two stages as well: torch's column reduction of a (1.65 M, N) matrix has only
            # N outputs to parallelise over (~200 GB/s, 3.1 ms per call in profiles/r1_h_train_iteration.txt;
            # rocBLAS gemv against a ones vector is worse: 16 ms); (G, R, N).sum(1) has G * N
            db = dyb.sum(1).sum(0) if T > 300000 else None
        if db is None:
            db = dy.sum(0)               # moderate row counts: one column reduction is fine
        elif Tp < T:
            db = db + dy[Tp:].sum(0)
        if Tp < T:
            dw = dw + dy[Tp:].t() @ x[Tp:]
        return (_dgrad(dy, weight) if need_dx else None), dw, (db if has_bias else None)


class _TallLinearReLU(torch.autograd.Function):
    """relu(x W^T + b) as one node (the FFN's first layer): the forward is one selfocc_linear_fwd launch with the ReLU in
    its epilogue; the backward masks dy with the saved output and continues as _TallLinear.  (nn.ReLU(inplace=True) on
    the view _TallLinear returns costs autograd a CopySlices: four extra 60 MB copies and a fill per layer.)"""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias):
        if FUSED_LINEAR_FWD and _linear_fwd_ok(x, weight):
            y = linear_fwd(x, weight, bias, relu=True)
        else:
            y = torch.relu(torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t())
        ctx.save_for_backward(x, weight, y)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = torch.ops.aten.threshold_backward(dy.contiguous().to(y.dtype), y, 0)
        return _tall_linear_backward(x, weight, dy, ctx.has_bias, ctx.needs_input_grad[0])


class _TallLinearHeads(torch.autograd.Function):
    """value_proj with a HEAD-MAJOR result under autograd: forward = selfocc_linear_fwd_heads (the projection writes the
    (G, B, 6, nv, 16) layout the MSDA kernels gather fastest from, forward and backward point kernels alike); backward
    brings the head-major gradient back to rows with one transposing copy and continues as _TallLinear."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias, nv, sink=None):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.sink = sink
        return linear_fwd_heads(x, weight, bias, nv)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, dy_hm):
        x, weight = ctx.saved_tensors
        G, B, H, nv, d = dy_hm.shape
        # the MSDA backward wrote its grad_value pixel-major into the sink (msda.ValueGradSink): dy_hm is a view of it
        dy = ctx.sink.rows([dy_hm[g] for g in range(G)]) if ctx.sink is not None else None
        if dy is None:
            dy = dy_hm.permute(1, 3, 0, 2, 4).reshape(B * nv, G * H * d)    # (b, pix, g, h, c): one transposing copy
        return (*_tall_linear_backward(x, weight, dy, ctx.has_bias, ctx.needs_input_grad[0]), None, None)


class _TallLinearHeadsMulti(torch.autograd.Function):
    """G value projections of the SAME rows (the TPV planes' value_proj of one layer): one selfocc_linear_fwd_heads launch
    with the stacked (G * 96, K) weight, one head-major tensor per group; backward = one weight-gradient and one
    input-gradient pass over the row-major (rows, G * 96) gradient assembled from the G head-major ones."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, nv, sink, *wb):
        w, b = stacked_view(list(wb[0::2])), stacked_view(list(wb[1::2]))
        if w is None or b is None:
            w, b = torch.cat(wb[0::2], 0), torch.cat(wb[1::2], 0)
        ctx.save_for_backward(x, w)
        ctx.G = len(wb) // 2
        ctx.sink = sink
        return tuple(linear_fwd_heads(x, w, b, nv).unbind(0))

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, *gys):
        x, w = ctx.saved_tensors
        G = ctx.G
        B, H, nv, d = next(g for g in gys if g is not None).shape
        # the G attentions' MSDA backward wrote their grad_value pixel-major into ONE row-major buffer (msda.ValueGradSink):
        # the gys are views of it — no transposing copies
        dy2 = ctx.sink.rows(list(gys)) if ctx.sink is not None else None
        if dy2 is None:
            dy = x.new_empty(B, nv, G, H, d)                       # (b, pix, g, h, c): row-major (rows, G * 96)
            for g, gy in enumerate(gys):
                if gy is None:
                    dy[:, :, g].zero_()
                else:
                    dy[:, :, g].copy_(gy.permute(0, 2, 1, 3))      # one transposing copy per group
            dy2 = dy.view(B * nv, G * H * d)
        dx, dw, db = _tall_linear_backward(x, w, dy2, True, ctx.needs_input_grad[0])
        n = H * d
        grads = [t for g in range(G) for t in (dw[g * n:(g + 1) * n], db[g * n:(g + 1) * n])]
        return (dx, None, None, *grads)


STACKED_VIEW = os.environ.get('SELFOCC_STACKED_VIEW', '1') == '1'      # A/B: 0 = torch.cat per call


def stacked_view(params):
    """ONE contiguous tensor that IS the row-concatenation of the leaf parameters ``params`` — their storages are made to
    alias consecutive row blocks of it — so that a stacked projection costs no ``torch.cat`` per call (the round-6 training
    iteration launched ~40 five-microsecond cats for the merged sampling_offsets | attention_weights and value projections).
    The parameters stay separate ``nn.Parameter`` objects with their own names, gradients and optimiser state; in-place
    updates (optimisers, ``load_state_dict``) keep the aliasing, a ``module.to(...)`` / ``.data =`` assignment breaks it and the
    next call re-establishes it.  None when a tensor is not a leaf parameter (``functional_call`` views under row sharding, ...)."""
    p0 = params[0]
    if not STACKED_VIEW:
        return None
    if any((not isinstance(p, nn.Parameter)) or (not p.is_leaf) or p.dtype != p0.dtype or p.device != p0.device
           or p.shape[1:] != p0.shape[1:] for p in params):
        return None
    rows = sum(p.shape[0] for p in params)
    inner = 1
    for n in p0.shape[1:]:
        inner *= n
    st, off, aliased = p0.untyped_storage(), p0.storage_offset(), True
    for p in params:
        if p.untyped_storage().data_ptr() != st.data_ptr() or p.storage_offset() != off or not p.is_contiguous():
            aliased = False
            break
        off += p.numel()
    if aliased:
        return torch.as_strided(p0.data, (rows, *p0.shape[1:]), p0.data.stride(), p0.storage_offset())
    with torch.no_grad():
        buf = torch.cat([p.data for p in params], 0).contiguous()
        o = 0
        for p in params:
            p.data = buf[o:o + p.shape[0]]
            o += p.shape[0]
    return buf


class _TallLinearMerged(torch.autograd.Function):
    """G Linears of the SAME rows as ONE projection with the weights stacked: y (T, N_1 + ... + N_G).  Used for the
    `sampling_offsets` | `attention_weights` pair of every deformable attention (image_cross_attention.py:296-312 and
    cross_view_hybrid_attention.py:78-90 read the same `query` twice): one read of x and one launch forward, and — because
    the MSDA backward writes the gradient of the merged row (ABI 32 ``ol_stride``) — ONE input-gradient pass, ONE
    weight-gradient pass and no gradient add in backward, instead of two each and an add."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, *wb):
        w, b = stacked_view(list(wb[0::2])), stacked_view(list(wb[1::2]))      # the parameters themselves, aliased: no copy
        if w is None or b is None:
            w, b = torch.cat(wb[0::2], 0), torch.cat(wb[1::2], 0)
        ctx.save_for_backward(x, w)
        ctx.sizes = [t.shape[0] for t in wb[0::2]]
        if FUSED_LINEAR_FWD and _linear_fwd_ok(x, w):
            return linear_fwd(x, w, b)
        return torch.addmm(b, x, w.t())

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = _tall_linear_backward(x, w, dy, True, ctx.needs_input_grad[0])
        grads, o = [], 0
        for n in ctx.sizes:
            grads += [dw[o:o + n], db[o:o + n]]
            o += n
        return (dx, *grads)


# sampling_offsets | attention_weights as one projection (env SELFOCC_MERGED_OFF_LOGITS=0: two Linears, as until round 5)
MERGED_OFF_LOGITS = os.environ.get('SELFOCC_MERGED_OFF_LOGITS', '1') == '1'
MERGED_OFF_LOGITS_CALLS = [0]      # calls served by the merged projection (tests read this)


def merged_off_logits(module, x2d):
    """(T, 3 * heads * L * P) rows [heads*L*P*2 raw offsets | heads*L*P logits] = the module's `sampling_offsets` and
    `attention_weights` Linears of the rows ``x2d`` (T, K) in ONE projection, or None when the module / input does not qualify
    (the caller then runs the two Linears).  Same parameters, same state-dict keys; under autograd the result carries one
    backward for both (``_TallLinearMerged``)."""
    so, aw = module.sampling_offsets, module.attention_weights
    if not (MERGED_OFF_LOGITS and x2d.is_cuda and x2d.dtype == torch.float32 and not torch.is_autocast_enabled()
            and so.bias is not None and aw.bias is not None and so.weight.dtype == torch.float32
            and so.weight.shape[0] == 2 * aw.weight.shape[0] and x2d.shape[0] >= LINEAR_FWD_MIN_ROWS):
        return None
    MERGED_OFF_LOGITS_CALLS[0] += 1
    if torch.is_grad_enabled() and (x2d.requires_grad or so.weight.requires_grad or aw.weight.requires_grad):
        return _TallLinearMerged.apply(x2d, so.weight, so.bias, aw.weight, aw.bias)
    w, b = stacked_view([so.weight, aw.weight]), stacked_view([so.bias, aw.bias])
    if w is None or b is None:
        w, b = torch.cat([so.weight, aw.weight], 0).detach(), torch.cat([so.bias, aw.bias], 0).detach()
    if FUSED_LINEAR_FWD and _linear_fwd_ok(x2d, w):
        return linear_fwd(x2d, w, b)
    return torch.addmm(b, x2d, w.t())


class TallLinear(nn.Linear):
    """nn.Linear (same parameters / state-dict keys) whose training forward goes through ``_TallLinear``
    when the input has many rows: the encoder's projections see 66 k - 153 k rows x 96 features, and the
    vendor GEMM chosen for their weight gradients (K = rows, 96 x 96 .. 96 x 2304 outputs) runs on a
    handful of workgroups (MT32x32x256: 0.3 ms x 32 calls per nuscenes_occ iteration)."""
    min_rows = 4096

    def forward(self, x):
        rows = x.numel() // max(x.shape[-1], 1)
        if torch.is_grad_enabled() and rows >= self.min_rows and (x.requires_grad or self.weight.requires_grad):
            y = _TallLinear.apply(x.reshape(rows, x.shape[-1]), self.weight, self.bias)
            return y.view(*x.shape[:-1], self.weight.shape[0])
        if not torch.is_grad_enabled() and FUSED_LINEAR_FWD and x.is_cuda and not torch.is_autocast_enabled():
            x2 = x.reshape(rows, x.shape[-1])
            if _linear_fwd_ok(x2, self.weight):
                return linear_fwd(x2, self.weight, self.bias).view(*x.shape[:-1], self.weight.shape[0])
        return super().forward(x)


@MODELS.register_module()
class FFN(BaseModule):
    """mmcv.cnn.bricks.transformer.FFN: Linear -> act -> drop (x num_fcs-1) -> Linear -> drop,
    returns identity + out."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2
        self.embed_dims, self.feedforward_channels, self.num_fcs = embed_dims, feedforward_channels, num_fcs
        layers, in_ch = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(TallLinear(in_ch, feedforward_channels), build_activation_layer(act_cfg),
                                        nn.Dropout(ffn_drop)))
            in_ch = feedforward_channels
        layers.append(TallLinear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None, post_norm=None):
        """``post_norm``: the layer's next `norm` module, applied to the result (TPVFormerLayer hands it over so that
        inference runs Linear + ReLU and Linear + identity + LayerNorm as two launches)."""
        if (not torch.is_grad_enabled() and self.num_fcs == 2 and self.add_identity and x.is_cuda
                and isinstance(self.layers[0][1], nn.ReLU) and isinstance(self.dropout_layer, nn.Identity)
                and not (self.training and (self.layers[0][2].p > 0 or self.layers[2].p > 0))):
            h = fused_linear(self.layers[0][0], x, relu=True)
            return fused_linear(self.layers[1], h, residual=x if identity is None else identity, norm=post_norm)
        out = self._forward_plain(x, identity)
        return post_norm(out) if post_norm is not None else out

    def _forward_plain(self, x, identity=None):
        lin0 = self.layers[0][0]
        rows = x.numel() // max(x.shape[-1], 1)
        if (FUSED_FFN_RELU and self.num_fcs == 2 and torch.is_grad_enabled() and x.is_cuda and isinstance(lin0, TallLinear)
                and isinstance(self.layers[0][1], nn.ReLU) and rows >= lin0.min_rows
                and (x.requires_grad or lin0.weight.requires_grad)):
            h = _TallLinearReLU.apply(x.reshape(rows, x.shape[-1]), lin0.weight, lin0.bias)
            h = self.layers[0][2](h.view(*x.shape[:-1], lin0.weight.shape[0]))
            out = self.layers[1](h)
            if FUSED_DROPOUT_ADD and self.add_identity and isinstance(self.dropout_layer, nn.Identity):
                # the last Dropout and the residual add as one pass (csrc/dropout.hip)
                drop = self.layers[2]
                return dropout_add(out, x if identity is None else identity, drop.p, drop.training)
            out = self.layers[2](out)
        else:
            out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


# training: the FFN's Linear + ReLU as one autograd node (_TallLinearReLU); env SELFOCC_FUSED_FFN_RELU=0: nn.Sequential
FUSED_FFN_RELU = os.environ.get('SELFOCC_FUSED_FFN_RELU', '1') == '1'
# training: `identity + dropout(x)` at the end of every attention / FFN block as one HIP pass per direction with a
# counter-based mask (selfocc_amd/dropout.py); env SELFOCC_FUSED_DROPOUT=0: torch's dropout + add
FUSED_DROPOUT_ADD = os.environ.get('SELFOCC_FUSED_DROPOUT', '1') == '1'
# training path of deformable_sampling: fused prologue + MSDA in both directions (msda.MSDAFusedFunction)
FUSED_TRAINING = True
# True: the fused / camera-loop kernels gather from a head-major copy of the projected value, (bs, heads, nv, d), where a
# cache line holds x-neighbours of one head.  Measured: camera-loop forward 0.87 -> 0.73 ms, fused
# forward 0.50 -> 0.47 ms, but the transposing copy per call costs more than that (eval encoder 12.5 -> 12.9 ms), so
# the default stays mmcv's (bs, nv, heads, d) as projected; the layout pays only if `value` is produced head-major.
HEAD_MAJOR_VALUE = False
# True: the projected `value` of the deformable attentions is STORED as bfloat16 for the gathers (forward and backward
# point kernels); arithmetic and gradients stay float32.  Halves the corner segments the L1-bound gathers move
# (BASELINE configs[1]: "bf16"); deviates from the reference's float32 MSDA by the bf16 rounding of value (~2^-9
# relative), so it is opt-in: the default reproduces the reference to 1e-4.  env SELFOCC_VALUE_BF16=1 sets it.
VALUE_BF16 = os.environ.get('SELFOCC_VALUE_BF16', '0') == '1'


def deformable_sampling(module, query, value, reference_points, spatial_shapes, level_start_index,
                        per_level_reference, key_padding_mask=None):
    """Shared body of the three deformable attentions: value_proj, offset / weight linears,
    softmax, sampling locations, MSDA (HIP).  ``per_level_reference``: reference points
    carry their own (level, point) dims (CrossViewHybridAttention,
    cross_view_hybrid_attention.py:96-99) instead of one anchor per point
    (BEVDeformableAttention, bevformer/attention/image_cross_attention.py:323-328)."""
    bs, num_query, _ = query.shape
    _, num_value, _ = value.shape
    host_shapes = getattr(spatial_shapes, '_so_host', None)
    if host_shapes is not None:      # no device read-back (a stream sync per call) when the caller knows the shapes
        assert sum(host_shapes[0::2][i] * host_shapes[1::2][i] for i in range(len(host_shapes) // 2)) == num_value
    else:
        assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == num_value
    LP0 = module.num_levels * module.num_points
    v_hm = None
    if (key_padding_mask is None and LP0 <= 256 and not HEAD_MAJOR_VALUE and module.value_proj.weight.shape[0] == 96
            and (not torch.is_grad_enabled() or FUSED_TRAINING)):
        # the projection itself writes (bs, heads, nv, d)
        v_hm = value_proj_head_major(module.value_proj.weight, module.value_proj.bias, value.reshape(bs * num_value, -1),
                                     num_value, module.num_heads)
    if v_hm is None:
        value = module.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, module.num_heads, -1)
    if reference_points.shape[-1] != 2:
        raise ValueError('Last dim of reference_points must be 2 on the SelfOcc path, '
                         f'got {reference_points.shape[-1]}')
    LP = module.num_levels * module.num_points
    d_head = module.value_proj.weight.shape[0] // module.num_heads
    use_bf16 = VALUE_BF16 and d_head == 16          # the bfloat16 gathers are built for 16 channels per head only
    kind = {'level': 0, 'point': 1, 'level_point': 2}[per_level_reference]
    fused_inf = not torch.is_grad_enabled() and LP <= 256 and msda_fused_kernels_built(d_head)
    fused_train = False
    if not fused_inf and FUSED_TRAINING and value.is_cuda and LP <= 256:
        host = getattr(spatial_shapes, '_so_host', None)
        if host is None:
            host = [int(v) for v in spatial_shapes.reshape(-1).tolist()]
        fused_train = msda_fused_supported(host, bs, num_query, module.num_heads, d_head, module.num_levels, module.num_points)
    # sampling_offsets | attention_weights: ONE projection with the stacked weight for the fused kernels (they read — and
    # in backward write — the merged row in place), the two Linears of the reference otherwise
    ol = merged_off_logits(module, query.reshape(bs * num_query, -1)) if (fused_inf or fused_train) else None
    if ol is not None:
        ol, off, logits, mlp = ol.view(bs, num_query, -1), None, None, (module.num_levels, module.num_points)
    else:
        mlp = None
        off = module.sampling_offsets(query).view(bs, num_query, module.num_heads, module.num_levels, module.num_points, 2)
    if fused_inf:
        # inference: softmax + sampling-location prologue fused into the HIP kernel (no loc / weight tensors)
        if ol is None:
            logits = module.attention_weights(query).view(bs, num_query, module.num_heads, LP)
        hm = HEAD_MAJOR_VALUE
        if v_hm is not None:
            value, hm = v_hm.view(v_hm.shape[1:]), True      # G = 1: a view (select's backward is a zero fill + a copy)
        elif HEAD_MAJOR_VALUE:
            value = to_head_major(value)
        if use_bf16:
            value = value.to(torch.bfloat16)
        return msda_fused_inference(value, spatial_shapes, level_start_index, reference_points, kind,
                                    ol if ol is not None else off, logits, hm, mlp)
    if fused_train:
        # training: the same fusion in both directions (no loc / weight tensors, no softmax / normalise kernels)
        if ol is None:
            logits = module.attention_weights(query).view(bs, num_query, module.num_heads, LP)
        hm = HEAD_MAJOR_VALUE
        if v_hm is not None:
            value, hm = v_hm.view(v_hm.shape[1:]), True      # G = 1: a view (select's backward is a zero fill + a copy)
        elif HEAD_MAJOR_VALUE:
            value = to_head_major(value)
        sink = getattr(v_hm, '_so_grad_sink', None) if v_hm is not None else None
        return MSDAFusedFunction.apply(value, spatial_shapes, level_start_index, reference_points, kind,
                                       ol if ol is not None else off, logits, host, hm, use_bf16, mlp,
                                       (sink, 0) if sink is not None else None)
    if v_hm is not None:      # (the unfused fallback below wants the mmcv layout)
        value = v_hm.view(v_hm.shape[1:]).permute(0, 2, 1, 3).contiguous()
    aw = module.attention_weights(query).view(bs, num_query, module.num_heads,
                                              module.num_levels * module.num_points).softmax(-1)
    aw = aw.view(bs, num_query, module.num_heads, module.num_levels, module.num_points)
    normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
    if per_level_reference == 'level_point':      # (bs, nq, L, P, 2)
        ref = reference_points[:, :, None, :, :, :]
    elif per_level_reference == 'point':          # (bs, nq, P, 2): one anchor per point, all levels
        ref = reference_points[:, :, None, None, :, :]
    else:                                         # (bs, nq, L, 2): mmcv base class
        ref = reference_points[:, :, None, :, None, :]
    loc = ref + off / normalizer[None, None, None, :, None, :]
    return MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index, loc, aw,
                                                  module.im2col_step)


@MODELS.register_module()
class MultiScaleDeformableAttention(BaseModule):
    """mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttention (mmcv==2.0.1): same
    constructor, parameters (sampling_offsets, attention_weights, value_proj, output_proj) and
    forward contract; used as self-attention by config/nuscenes/nuscenes_occ_bev.py:222."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None, value_proj_ratio=1.0):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}')
        dim_per_head = embed_dims // num_heads
        if dim_per_head & (dim_per_head - 1):
            warnings.warn("the HIP MSDA kernels need a power-of-two dim per head: 8, 16 or 32 for the fused / camera-loop ops, "
                          "4 on the plain op only")
        self.norm_cfg, self.batch_first = norm_cfg, batch_first
        self.dropout = nn.Dropout(dropout)
        self.im2col_step, self.embed_dims = im2col_step, embed_dims
        self.num_levels, self.num_heads, self.num_points = num_levels, num_heads, num_points
        self.sampling_offsets = TallLinear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = TallLinear(embed_dims, num_heads * num_levels * num_points)
        value_proj_size = int(embed_dims * value_proj_ratio)
        self.value_proj = TallLinear(embed_dims, value_proj_size)
        self.output_proj = TallLinear(value_proj_size, embed_dims)
        self.init_weights()

    _scale_points = True  # mmcv scales the i-th point's offset bias by (i + 1)

    def init_weights(self):
        constant_init(self.sampling_offsets, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        if self._scale_points:
            for i in range(self.num_points):
                grid_init[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = grid_init.view(-1).to(self.sampling_offsets.bias.device)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        if hasattr(self, 'output_proj'):
            xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    _reference_kind = 'level'

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        output = deformable_sampling(self, query, value, reference_points, spatial_shapes, level_start_index,
                                     self._reference_kind, key_padding_mask)
        post_norm = kwargs.get('post_norm')
        if not torch.is_grad_enabled() and not self.training and self.batch_first and output.is_cuda:
            return fused_linear(self.output_proj, output, residual=identity, norm=post_norm)
        output = self.output_proj(output)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        if FUSED_DROPOUT_ADD:
            output = dropout_add(output, identity, self.dropout.p, self.dropout.training)
        else:
            output = self.dropout(output) + identity
        return post_norm(output) if post_norm is not None else output
