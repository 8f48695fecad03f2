"""TPVFormer / BEVFormer encoders: orchestration around the two MSDA call sites.

TPVPositionalEncoding <- model/encoder/tpvformer/tpvformer_pos_embed.py:6-57
BEVPositionalEncoding <- model/encoder/bevformer/bevformer_pos_embed.py:7-34
TPVFormerLayer        <- model/encoder/tpvformer/tpvformer_encoder_layer.py:11-219
BEVFormerLayer        <- model/encoder/bevformer/bevformer_encoder_layer.py:11-216
TPVFormerEncoder      <- model/encoder/tpvformer/tpvformer_encoder.py:20-290
BEVFormerEncoder      <- model/encoder/bevformer/bevformer_encoder.py:18-224
Same registry names, constructor kwargs, buffers / parameter names and forward contracts.
"""
import copy
import os

import torch
import torch.nn as nn

from ...mapping import GridMeterMapping
from ...registry import (MODELS, build_attention, build_feedforward_network, build_positional_encoding,
                         build_transformer_layer)
from ..bricks import BaseModule, ModuleList, MultiScaleDeformableAttention, TallLinear, build_norm_layer
from .attention import BEVCrossAttention, BEVDeformableAttention, TPVCrossAttention, CrossViewHybridAttention
from .utils import point_sampling, get_cross_view_ref_points


def _fourier(num_freqs, meter):
    """(…, 2) normalised metres -> (N, 4 * num_freqs) sin/cos features, pi * 2^k, k = -1 .. F-2."""
    freqs = torch.pi * (2 ** torch.arange(-1, num_freqs - 1, dtype=torch.float))
    mf = meter.unsqueeze(-1) * freqs[None, None, None, ...]
    return torch.stack([torch.sin(mf), torch.cos(mf)], dim=-1).flatten(-3).flatten(0, 1)


def _normalise(meter, lo0, hi0, lo1, hi1):
    meter = meter.clone()
    meter[..., 0] = (meter[..., 0] - lo0) / (hi0 - lo0)
    meter[..., 1] = (meter[..., 1] - lo1) / (hi1 - lo1)
    return meter


_PLANE_SHAPES = {}


def _plane_shapes(H, W, Z, device):
    """(spatial_shapes, level_start_index) of the three TPV planes as 'levels' of the cross-view attention;
    cached per (size, device) — a host list -> device tensor copy is a synchronising call."""
    key = (H, W, Z, str(device))
    hit = _PLANE_SHAPES.get(key)
    if hit is None:
        ss = torch.tensor([[H, W], [Z, H], [W, Z]], device=device)
        ss._so_host = [H, W, Z, H, W, Z]
        lsi = torch.tensor([0, H * W, H * W + Z * H], device=device)
        hit = _PLANE_SHAPES[key] = (ss, lsi)
    return hit


_LEVEL_SHAPES = {}


def _level_shapes(shapes, device):
    """(spatial_shapes, level_start_index) of the FPN levels, cached per (shapes, device): the same every frame."""
    key = (shapes, str(device))
    hit = _LEVEL_SHAPES.get(key)
    if hit is None:
        ss = torch.as_tensor(shapes, dtype=torch.long, device=device)
        ss._so_host = [int(v) for hw in shapes for v in hw]   # host copy for the MSDA backward decomposition
        lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
        hit = _LEVEL_SHAPES[key] = (ss, lsi)
    return hit


@MODELS.register_module()
class TPVPositionalEncoding(BaseModule):
    def __init__(self, num_freqs, embed_dims, tpv_meters, tot_range, init_cfg=None):
        super().__init__(init_cfg)
        assert isinstance(tot_range, list) and len(tot_range) == 6
        r = tot_range
        hw, zh, wz = tpv_meters
        self.register_buffer('hw_freq_feat', _fourier(num_freqs[0], _normalise(hw, r[0], r[3], r[1], r[4])), False)
        self.register_buffer('zh_freq_feat', _fourier(num_freqs[1], _normalise(zh, r[1], r[4], r[2], r[5])), False)
        self.register_buffer('wz_freq_feat', _fourier(num_freqs[2], _normalise(wz, r[0], r[3], r[2], r[5])), False)
        # TallLinear (= nn.Linear, same keys): the weight gradient over 66 k rows with a 96 x 48 result took the vendor
        # GEMM 0.4 ms (one workgroup column); the row-split form of bricks._tall_linear_backward takes ~40 us
        self.position_layer_hw = TallLinear(4 * num_freqs[0], embed_dims)
        self.position_layer_zh = TallLinear(4 * num_freqs[1], embed_dims)
        self.position_layer_wz = TallLinear(4 * num_freqs[2], embed_dims)

    def forward(self):
        return [self.position_layer_hw(self.hw_freq_feat), self.position_layer_zh(self.zh_freq_feat),
                self.position_layer_wz(self.wz_freq_feat)]


@MODELS.register_module()
class BEVPositionalEncoding(BaseModule):
    def __init__(self, num_freqs, embed_dims, bev_meter, tot_range, init_cfg=None):
        super().__init__(init_cfg)
        r = tot_range if isinstance(tot_range, list) else [-1.0 * tot_range, -1.0 * tot_range, 0., tot_range, tot_range, 0.]
        self.register_buffer('freq_feat', _fourier(num_freqs, _normalise(bev_meter, r[0], r[3], r[1], r[4])), False)
        self.position_layer = TallLinear(4 * num_freqs, embed_dims)

    def forward(self):
        return self.position_layer(self.freq_feat)


def cat_planes(planes):
    """torch.cat(planes, dim=1) — without the copy when the planes already ARE consecutive slices of one
    contiguous (1, N, C) buffer (the views torch.split hands out, or the slices TPVCrossAttention writes into):
    the encoder layer alternates between the concatenated and the per-plane form five times, 30 MB per copy at the
    shipped sizes.  Under autograd the plain cat is kept (its backward is already a view)."""
    if torch.is_grad_enabled() or len(planes) == 1:
        return planes[0] if len(planes) == 1 else torch.cat(planes, dim=1)
    first = planes[0]
    C = first.shape[-1]
    ok = first.dim() == 3 and first.shape[0] == 1
    off = first.storage_offset()
    for p in planes:
        ok = ok and p.dim() == 3 and p.shape[0] == 1 and p.shape[-1] == C and p.stride(1) == C and p.stride(2) == 1 \
            and p.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() and p.storage_offset() == off \
            and p.dtype == first.dtype
        off += p.shape[1] * C
    if not ok:
        return torch.cat(planes, dim=1)
    n = sum(p.shape[1] for p in planes)
    return first.as_strided((1, n, C), (n * C, C, 1), first.storage_offset())


class _Planes(tuple):
    """The three TPV planes as the torch.split views of ONE concatenated (1, N, C) tensor, which is kept (``cat``): the
    encoder layer alternates between the per-plane form (image cross-attention) and the concatenated form (cross-view
    self-attention, norm, ffn), and under autograd ``torch.cat(torch.split(t))`` is a 30 MB copy forward plus another one
    backward, six times per layer.  Taking ``cat`` back is the same function of ``t`` without either copy."""
    cat = None


def _as_cat(q):
    if torch.is_tensor(q):
        return q
    c = getattr(q, 'cat', None)
    return c if c is not None else cat_planes(list(q))


def _as_planes(q, sizes):
    if not torch.is_tensor(q):
        return q
    p = _Planes(torch.split(q, sizes, 1))
    p.cat = q
    return p


class _FormerLayerBase(BaseModule):
    """attention / norm / ffn stack driven by ``operation_order`` (mmcv BaseTransformerLayer idiom)."""

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type='FFN', feedforward_channels=1024, num_fcs=2, ffn_drop=0.,
                               act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None, batch_first=True, **kwargs):
        ffn_cfgs = copy.deepcopy(ffn_cfgs)
        for old, new in dict(feedforward_channels='feedforward_channels', ffn_dropout='ffn_drop',
                             ffn_num_fcs='num_fcs').items():
            if old in kwargs:
                ffn_cfgs[new] = kwargs[old]
        super().__init__(init_cfg)
        self.batch_first = batch_first
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            attn_cfgs = [copy.deepcopy(c) for c in attn_cfgs]
            assert num_attn == len(attn_cfgs)
        self.num_attn, self.operation_order, self.norm_cfg = num_attn, operation_order, norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = ModuleList()
        index = 0
        for op in operation_order:
            if op in ('self_attn', 'cross_attn'):
                if 'batch_first' in attn_cfgs[index]:
                    assert self.batch_first == attn_cfgs[index]['batch_first']
                else:
                    attn_cfgs[index]['batch_first'] = self.batch_first
                attention = build_attention(attn_cfgs[index])
                attention.operation_name = op
                self.attentions.append(attention)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = ModuleList()
        num_ffns = operation_order.count('ffn')
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [copy.deepcopy(ffn_cfgs) for _ in range(num_ffns)]
        assert len(ffn_cfgs) == num_ffns
        for cfg in ffn_cfgs:
            cfg.setdefault('embed_dims', self.embed_dims)
            assert cfg['embed_dims'] == self.embed_dims
            self.ffns.append(build_feedforward_network(cfg))
        self.norms = ModuleList([build_norm_layer(norm_cfg, self.embed_dims)[1]
                                 for _ in range(operation_order.count('norm'))])


@MODELS.register_module()
class TPVFormerLayer(_FormerLayerBase):
    def __init__(self, *args, multi_plane_ffn_norm=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.multi_plane_ffn_norm = multi_plane_ffn_norm

    def forward(self, query, key=None, value=None, tpv_pos=None, ref_2d=None, spatial_shapes=None,
                level_start_index=None, reference_points_cams=None, tpv_masks=None, tpv_size=None, **kwargs):
        H, W, Z = tpv_size
        # row-sharded encoder (TPVFormerEncoder(row_shard=True)): `query` holds this rank's rows of every plane
        # (`plane_sizes`), the cross-view self-attention still samples ALL rows (`self_attn_value`, replicated); every other
        # step is row-wise
        sizes = kwargs.pop('plane_sizes', None) or [H * W, Z * H, W * Z]
        self_value = kwargs.pop('self_attn_value', None)
        tpv_pos_cat = kwargs.pop('tpv_pos_cat', None)
        if tpv_pos_cat is None:
            tpv_pos_cat = torch.cat(tpv_pos, dim=1)
        norm_i = attn_i = ffn_i = 0
        identity = query
        device = query[0].device
        # `query` / `identity` are a concatenated tensor or the planes (see _Planes); each step takes the form it needs
        mp = self.multi_plane_ffn_norm
        cat = (lambda q: tuple(_as_planes(q, sizes))) if mp else _as_cat
        # inference: a `norm` that directly follows an attention / ffn step rides in that step's last projection
        # (selfocc_linear_fwd: output_proj + residual + LayerNorm in one launch) instead of re-reading the planes
        # (post-norm layers only: with pre_norm the un-normalised tensor is still needed as the next identity)
        fuse_norm = (not torch.is_grad_enabled() and not self.training and not mp
                     and not self.pre_norm and query[0].is_cuda)
        skip_norm = False
        ops = self.operation_order
        for k, op in enumerate(ops):
            post_norm = None
            if fuse_norm and op != 'norm' and k + 1 < len(ops) and ops[k + 1] == 'norm':
                post_norm = self.norms[norm_i]
                skip_norm = True
            if op == 'self_attn':   # cross-view hybrid attention: the 3 planes are the 3 "levels"
                ss, lsi = _plane_shapes(H, W, Z, device)     # constants: uploaded once, not once per layer call
                q = _as_cat(query)
                v = q if self_value is None else self_value
                query = self.attentions[attn_i](q, v, v, _as_cat(identity) if self.pre_norm else None,
                                                query_pos=tpv_pos_cat, reference_points=ref_2d,
                                                spatial_shapes=ss, level_start_index=lsi, post_norm=post_norm, **kwargs)
                attn_i += 1
                identity = query
            elif op == 'norm':
                if skip_norm:       # already applied inside the previous step
                    skip_norm = False
                else:
                    query = self.norms[norm_i](cat(query))
                norm_i += 1
            elif op == 'cross_attn':  # image cross-attention, per plane
                query = self.attentions[attn_i](_as_planes(query, sizes), key, value,
                                                _as_planes(identity, sizes) if self.pre_norm else None,
                                                spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                                reference_points_cams=reference_points_cams, tpv_masks=tpv_masks,
                                                post_norm=post_norm, **kwargs)
                attn_i += 1
                identity = query
            elif op == 'ffn':
                query = self.ffns[ffn_i](cat(query), cat(identity) if self.pre_norm else None, post_norm=post_norm)
                ffn_i += 1
        return _as_planes(query, sizes)


@MODELS.register_module()
class BEVFormerLayer(_FormerLayerBase):
    def forward(self, query, key=None, value=None, bev_pos=None, ref_2d=None, spatial_shapes=None,
                level_start_index=None, reference_points_cams=None, bev_masks=None, bev_size=None, **kwargs):
        norm_i = attn_i = ffn_i = 0
        identity = query
        self_value = kwargs.pop('self_attn_value', None)   # row-sharded encoder: `query` = this rank's rows, the self-attention samples ALL rows
        # inference, post-norm layers: a `norm` that follows an attention / ffn step rides in that step's last
        # projection (selfocc_linear_fwd), as in TPVFormerLayer
        fuse_norm = not torch.is_grad_enabled() and not self.training and not self.pre_norm and query.is_cuda
        skip_norm = False
        ops = self.operation_order
        for k, op in enumerate(ops):
            post_norm = None
            if fuse_norm and op != 'norm' and k + 1 < len(ops) and ops[k + 1] == 'norm':
                post_norm = self.norms[norm_i]
                skip_norm = True
            if op == 'self_attn':
                ss, lsi = _level_shapes((tuple(int(v) for v in bev_size),), query.device)   # cached: no upload per call
                sv = query if self_value is None else self_value
                query = self.attentions[attn_i](query, sv, sv, identity if self.pre_norm else None,
                                                query_pos=bev_pos, reference_points=ref_2d, spatial_shapes=ss,
                                                level_start_index=lsi, post_norm=post_norm, **kwargs)
                attn_i += 1
                identity = query
            elif op == 'norm':
                if skip_norm:
                    skip_norm = False
                else:
                    query = self.norms[norm_i](query)
                norm_i += 1
            elif op == 'cross_attn':
                query = self.attentions[attn_i](query, key, value, identity if self.pre_norm else None,
                                                spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                                reference_points_cams=reference_points_cams, bev_masks=bev_masks,
                                                post_norm=post_norm, **kwargs)
                attn_i += 1
                identity = query
            elif op == 'ffn':
                query = self.ffns[ffn_i](query, identity if self.pre_norm else None, post_norm=post_norm)
                ffn_i += 1
        return query


class _EncoderBase(BaseModule):
    _attention_types = (BEVCrossAttention, MultiScaleDeformableAttention, BEVDeformableAttention,
                        TPVCrossAttention, CrossViewHybridAttention)

    def _build_layers(self, transformerlayers, num_layers):
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, (list, tuple)) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = ModuleList([build_transformer_layer(copy.deepcopy(c)) for c in transformerlayers])
        self.pre_norm = self.layers[0].pre_norm
        self.level_embeds = nn.Parameter(torch.randn(self.num_feature_levels, self.embed_dims))
        self.cams_embeds = nn.Parameter(torch.randn(self.num_cams, self.embed_dims))

    row_shard = False

    def _row_sharding(self):
        """True when this call splits the plane rows over the ranks: ``row_shard=True`` (or SELFOCC_ENC_SHARD=1) and an
        initialised process group with world_size > 1.  Like NeuSHead(ray_shard=True) it assumes that every rank holds
        the SAME frame (one frame's work split over the ranks)."""
        import torch.distributed as tdist
        on = self.row_shard or os.environ.get('SELFOCC_ENC_SHARD', '0') == '1'
        return bool(on and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1)

    def _row_shard_plan_for(self, sizes):
        from ... import dist as sdist
        shard = getattr(self, '_row_shard_plan', None)
        if shard is None or shard.sizes != list(sizes) or (shard.rank, shard.world_size) != sdist.world():
            shard = self._row_shard_plan = sdist.PlaneRowShard(list(sizes))
        return shard

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, self._attention_types):
                (getattr(m, 'init_weight', None) or m.init_weights)()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.cams_embeds)

    def _positions(self, bs):
        """The positional encodings (a Linear over constant Fourier features), batched, and — for several planes — their
        concatenation.  Without autograd they depend on the layers' parameters only: computed once and kept until a
        parameter changes (0.12 ms per TPV frame: three small GEMMs, their batch copies and a cat)."""
        pe = self.positional_encoding

        def build():
            out = pe()
            if isinstance(out, (list, tuple)):
                pos = [p.unsqueeze(0).expand(bs, -1, -1) for p in out]
                return pos, (None if torch.is_grad_enabled() else torch.cat(pos, dim=1))
            return out.unsqueeze(0).expand(bs, -1, -1), None
        if torch.is_grad_enabled():
            return build()
        key = (bs,) + tuple((None if p.is_inference() else p._version, p.data_ptr()) for p in pe.parameters())
        hit = getattr(self, '_pos_cache', None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = self._pos_cache = (key, build())
        return hit[1]

    def _flatten_feats(self, img_feats):
        """4 x (B, N, C, h, w) -> (N, sum hw, B, C) + cam / level embeddings, spatial shapes, level starts."""
        device = img_feats[0].device
        shapes = tuple((f.shape[3], f.shape[4]) for f in img_feats)
        if img_feats[0].is_cuda and torch.is_grad_enabled() and FLATTEN_FEATS_FUNCTION:
            flat = _FlattenFeats.apply(self.cams_embeds, self.level_embeds, *img_feats)
        elif torch.is_grad_enabled() and (self.cams_embeds.requires_grad or any(f.requires_grad for f in img_feats)):
            flat = _flatten_feats_torch(self.cams_embeds, self.level_embeds, img_feats)     # autograd through the torch ops
        else:
            flat = _flatten_feats(self.cams_embeds, self.level_embeds, img_feats)
        spatial_shapes, level_start_index = _level_shapes(shapes, device)
        return flat, spatial_shapes, level_start_index


# the flatten as one HIP pass (csrc/geometry.hip, selfocc_flatten_feats) instead of 2 adds per level + cat; env
# SELFOCC_FLATTEN_HIP=0: the torch ops
FLATTEN_HIP = os.environ.get('SELFOCC_FLATTEN_HIP', '1') == '1'


def _flatten_feats(cams_embeds, level_embeds, img_feats):
    f0 = img_feats[0]
    if (FLATTEN_HIP and f0.is_cuda and not torch.is_autocast_enabled() and 1 <= len(img_feats) <= 8 and f0.shape[2] <= 512
            and all(f.dtype == torch.float32 and f.dim() == 5 and f.shape[:3] == f0.shape[:3] for f in img_feats)
            and cams_embeds.dtype == torch.float32 and level_embeds.dtype == torch.float32
            and cams_embeds.shape[0] == f0.shape[1] and level_embeds.shape[0] >= len(img_feats)):
        import ctypes as C
        from ..._lib import lib, check, ptr, current_stream
        B, N, Cc = f0.shape[:3]
        feats = [f.contiguous() for f in img_feats]
        hw = [f.shape[3] * f.shape[4] for f in feats]
        out = f0.new_empty(N, sum(hw), B, Cc)
        ptrs = (C.c_void_p * len(feats))(*[f.data_ptr() for f in feats])
        check(lib().selfocc_flatten_feats(ptrs, (C.c_int32 * len(hw))(*hw), len(feats), B, N, Cc,
                                          ptr(cams_embeds.detach().contiguous()), ptr(level_embeds.detach().contiguous()),
                                          ptr(out), current_stream(f0.device)), "selfocc_flatten_feats")
        return out
    return _flatten_feats_torch(cams_embeds, level_embeds, img_feats)


def _flatten_feats_torch(cams_embeds, level_embeds, img_feats):
    flat = []
    for lvl, feat in enumerate(img_feats):
        feat = feat.flatten(3).permute(1, 0, 3, 2)           # N, B, hw, C
        feat = feat + cams_embeds[:, None, None, :].to(feat.dtype)
        feat = feat + level_embeds[None, None, lvl:lvl + 1, :].to(feat.dtype)
        flat.append(feat)
    return torch.cat(flat, 2).permute(0, 2, 1, 3)


# training: the embedding gradients of _flatten_feats through column sums in two stages (env SELFOCC_FLATTEN_FEATS_FN=0:
# torch autograd, whose reduction of a (6, 1, 19 200, 96) gradient to (6, 1, 1, 96) takes 0.22 ms, twice per level)
FLATTEN_FEATS_FUNCTION = os.environ.get('SELFOCC_FLATTEN_FEATS_FN', '1') == '1'


def _colsum(t):
    """(N, R, C) -> (N, C) in two stages: the first has N * R / 128 * C outputs to parallelise over."""
    N, R, C = t.shape
    k = 128
    main = R // k * k
    out = t.new_zeros(N, C)
    if main:
        out = out + t[:, :main].reshape(N, main // k, k, C).sum(2).sum(1)
    if main < R:
        out = out + t[:, main:].sum(1)
    return out


class _FlattenFeats(torch.autograd.Function):
    """_flatten_feats with a hand-written backward: d cams_embeds[n] / d level_embeds[l] are column sums of the flat
    gradient over (level slice, batch) — computed once per (camera, level) and combined — and the image-feature gradient
    is the slice itself, transposed back."""

    @staticmethod
    def forward(ctx, cams_embeds, level_embeds, *img_feats):
        ctx.shapes = [tuple(f.shape) for f in img_feats]
        ctx.dtypes = (cams_embeds.dtype, level_embeds.dtype)
        ctx.n_levels = level_embeds.shape[0]
        return _flatten_feats(cams_embeds, level_embeds, img_feats)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):                                    # (N, S, B, C)
        N, S, B, C = g.shape
        g = g.contiguous()
        per, feats, s0 = [], [], 0
        for i, (b, n, c, h, w) in enumerate(ctx.shapes):
            seg = g[:, s0:s0 + h * w]                        # (N, hw, B, C)
            s0 += h * w
            per.append(_colsum(seg.reshape(N, h * w * B, C).float()))          # (N, C) of this level
            feats.append(seg.permute(2, 0, 3, 1).reshape(b, n, c, h, w) if ctx.needs_input_grad[2 + i] else None)
        per = torch.stack(per)                               # (L, N, C)
        g_cam = per.sum(0).to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None
        g_lvl = None
        if ctx.needs_input_grad[1]:
            g_lvl = per.new_zeros(ctx.n_levels, C)           # levels beyond the maps handed in keep a zero gradient
            g_lvl[:per.shape[0]] = per.sum(1)
            g_lvl = g_lvl.to(ctx.dtypes[1])
        return (g_cam, g_lvl, *feats)


@MODELS.register_module()
class TPVFormerEncoder(_EncoderBase):
    def __init__(self, mapping_args, embed_dims=128, num_cams=6, num_feature_levels=4, positional_encoding=None,
                 num_points_cross=[64, 64, 8], num_points_self=[16, 16, 16], transformerlayers=None,
                 num_layers=None, camera_aware=False, camera_aware_mid_channels=None, init_cfg=None, row_shard=False):
        super().__init__(init_cfg)
        self.row_shard = row_shard          # split the plane rows over the ranks (one frame on all ranks; dist.PlaneRowShard)
        if camera_aware:
            raise NotImplementedError("camera_aware=True (CameraAwareSE) is off in every shipped config")
        self.embed_dims, self.num_feature_levels, self.num_cams = embed_dims, num_feature_levels, num_cams
        self.camera_aware = camera_aware
        self.mapping = GridMeterMapping(**mapping_args)
        H, W, Z = self.mapping.size_h, self.mapping.size_w, self.mapping.size_d
        ar = lambda n: torch.arange(n, dtype=torch.float)
        hw_grid = torch.stack([ar(H)[:, None].expand(-1, W), ar(W)[None].expand(H, -1), torch.zeros(H, W)], -1)
        zh_grid = torch.stack([ar(H)[None].expand(Z, -1), torch.zeros(Z, H), ar(Z)[:, None].expand(-1, H)], -1)
        wz_grid = torch.stack([torch.zeros(W, Z), ar(W)[:, None].expand(-1, Z), ar(Z)[None].expand(W, -1)], -1)
        g2m = self.mapping.grid2meter
        positional_encoding = dict(positional_encoding)
        positional_encoding['tpv_meters'] = [g2m(hw_grid)[..., [0, 1]], g2m(zh_grid)[..., [1, 2]], g2m(wz_grid)[..., [0, 2]]]
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.tpv_size = [H, W, Z]
        self._build_layers(transformerlayers, num_layers)
        self.num_points_cross, self.num_points_self = num_points_cross, num_points_self

        # pillar reference points (metres) of the image cross-attention, one pillar per plane cell
        Pz, Pw, Ph = num_points_cross[2], num_points_cross[1], num_points_cross[0]
        hw3 = torch.cat([hw_grid[..., [0, 1]].unsqueeze(2).expand(-1, -1, Pz, -1),
                         torch.linspace(0, Z - 1, Pz).reshape(1, 1, -1, 1).expand(H, W, -1, -1)], -1)
        zh3 = torch.cat([zh_grid[..., :1].unsqueeze(2).expand(-1, -1, Pw, -1),
                         torch.linspace(0, W - 1, Pw).reshape(1, 1, -1, 1).expand(Z, H, -1, -1),
                         zh_grid[..., 2:].unsqueeze(2).expand(-1, -1, Pw, -1)], -1)
        wz3 = torch.cat([torch.linspace(0, H - 1, Ph).reshape(1, 1, -1, 1).expand(W, Z, -1, -1),
                         wz_grid[..., [1, 2]].unsqueeze(2).expand(-1, -1, Ph, -1)], -1)
        for name, g in (('ref_3d_hw', hw3), ('ref_3d_zh', zh3), ('ref_3d_wz', wz3)):
            self.register_buffer(name, g2m(g).flatten(0, 1).transpose(0, 1).contiguous(), False)   # (D, Q, 3) as point_sampling reads it
        self.register_buffer('cross_view_ref_points', get_cross_view_ref_points(H, W, Z, num_points_self), False)

    def _forward_layers_sharded(self, tpv_query, key, value, tpv_pos, tpv_pos_cat, spatial_shapes, level_start_index,
                                reference_points_cams, tpv_masks, ref_cross_view, **kwargs):
        """every rank owns a row block of each plane.  Per layer: cross-view self-attention of the local
        rows over the full (replicated) planes, image cross-attention / norms / FFN on the local rows, then ONE all-gather
        of the updated rows (78 899 x 96 floats = 30 MB at the shipped size).  Under autograd the gather's backward sums
        the ranks' partial gradients of the gathered planes, and each layer's parameter gradients (computed from the local
        rows only) are summed in one coalesced all-reduce per layer, so that every rank ends up with the gradients of the
        unsharded encoder; the image features' and the incoming queries' gradients likewise (replicate_grad_sum)."""
        from ... import dist as sdist
        from torch.func import functional_call
        H, W, Z = self.tpv_size
        sizes = [H * W, Z * H, W * Z]
        if tpv_query[0].shape[0] != 1:
            raise NotImplementedError("row_shard=True splits ONE frame over the ranks: batch size must be 1 (got "
                                      f"{tpv_query[0].shape[0]}), as everywhere in the reference's head (neus_head.py:523)")
        shard = self._row_shard_plan_for(sizes)
        grad = torch.is_grad_enabled()
        q_full = _as_cat(tpv_query)
        if grad:
            q_full = sdist.replicate_grad_sum(q_full)                    # the lifter's queries: used by local rows only
            key = value = sdist.replicate_grad_sum(value)                # image features: sampled by local queries only
            tpv_pos_cat = sdist.replicate_grad_sum(torch.cat(tpv_pos, dim=1) if tpv_pos_cat is None else tpv_pos_cat)
        elif tpv_pos_cat is None:
            tpv_pos_cat = torch.cat(tpv_pos, dim=1)
        pos_loc = shard.take(tpv_pos_cat, 1)
        pos_planes = list(torch.split(pos_loc, shard.local_sizes, 1))
        ref_loc = shard.take(ref_cross_view, 1)
        cams_loc = [shard.take_plane(c, i, 2) for i, c in enumerate(reference_points_cams)]
        masks_loc = [shard.take_plane(m, i, 2) for i, m in enumerate(tpv_masks)]
        last = len(self.layers) - 1
        for li, layer in enumerate(self.layers):
            q_loc = _as_planes(shard.take(q_full, 1), shard.local_sizes)
            call = dict(tpv_pos=pos_planes, tpv_pos_cat=pos_loc, ref_2d=ref_loc, spatial_shapes=spatial_shapes,
                        level_start_index=level_start_index, reference_points_cams=cams_loc, tpv_masks=masks_loc,
                        tpv_size=self.tpv_size, rebatch_plans=None, plane_sizes=shard.local_sizes, self_attn_value=q_full,
                        **kwargs)
            if grad:
                names, ps = zip(*layer.named_parameters())
                out = functional_call(layer, dict(zip(names, sdist.group_grad_sum(ps))), (q_loc, key, value), call)
            else:
                out = layer(q_loc, key, value, **call)
            q_full = sdist.gather_plane_rows(_as_cat(out), shard, reduce_grad=li < last)
        return _as_planes(q_full, sizes)

    def forward_layers(self, tpv_query, key, value, tpv_pos=None, spatial_shapes=None, level_start_index=None,
                       img_metas=None, **kwargs):
        bs = tpv_query[0].shape[0]
        reference_points_cams, tpv_masks = [], []
        for ref_3d in (self.ref_3d_hw, self.ref_3d_zh, self.ref_3d_wz):
            cam, mask = point_sampling(ref_3d.unsqueeze(0).expand(bs, -1, -1, -1), img_metas)   # read-only: no copy at bs = 1
            reference_points_cams.append(cam)
            tpv_masks.append(mask)
        ref_cross_view = self.cross_view_ref_points.unsqueeze(0).expand(bs, -1, -1, -1, -1)      # read-only downstream
        # the camera-loop kernels need no re-batch plan (no host sync); a layer that cannot take that path
        # (batch > 1, shapes the banded scatter does not cover) builds its own
        plans = None
        tpv_pos_cat = kwargs.pop('tpv_pos_cat', None)
        if self._row_sharding():
            return self._forward_layers_sharded(tpv_query, key, value, tpv_pos, tpv_pos_cat, spatial_shapes, level_start_index,
                                                reference_points_cams, tpv_masks, ref_cross_view, **kwargs)
        if tpv_pos_cat is None:
            tpv_pos_cat = torch.cat(tpv_pos, dim=1)    # once per forward, not once per layer
        for layer in self.layers:
            tpv_query = layer(tpv_query, key, value, tpv_pos=tpv_pos, tpv_pos_cat=tpv_pos_cat, ref_2d=ref_cross_view,
                              spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                              reference_points_cams=reference_points_cams, tpv_masks=tpv_masks,
                              tpv_size=self.tpv_size, rebatch_plans=plans, **kwargs)
        return tpv_query

    def forward(self, representation, ms_img_feats=None, metas=None, **kwargs):
        bs = ms_img_feats[0].shape[0]
        tpv_pos, tpv_pos_cat = self._positions(bs)
        feat, spatial_shapes, level_start_index = self._flatten_feats(ms_img_feats)
        tpv = self.forward_layers(representation, feat, feat, tpv_pos=tpv_pos, spatial_shapes=spatial_shapes,
                                  level_start_index=level_start_index, img_metas=metas, tpv_pos_cat=tpv_pos_cat)
        return {'representation': tpv}


@MODELS.register_module()
class BEVFormerEncoder(_EncoderBase):
    def __init__(self, mapping_args, embed_dims=128, num_cams=6, num_feature_levels=4, positional_encoding=None,
                 num_points_cross=32, num_points_self=16, transformerlayers=None, num_layers=None, init_cfg=None,
                 row_shard=False):
        super().__init__(init_cfg)
        self.row_shard = row_shard          # split the BEV rows over the ranks (one frame on all ranks; dist.PlaneRowShard)
        self.embed_dims, self.num_feature_levels, self.num_cams = embed_dims, num_feature_levels, num_cams
        self.mapping = GridMeterMapping(**mapping_args)
        H, W, Z = self.mapping.size_h, self.mapping.size_w, self.mapping.size_d
        ar = lambda n: torch.arange(n, dtype=torch.float)
        bev_grid = torch.stack([ar(H)[:, None].expand(-1, W), ar(W)[None].expand(H, -1)], -1)
        positional_encoding = dict(positional_encoding)
        positional_encoding['bev_meter'] = self.mapping.grid2meter(bev_grid)
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.bev_size = [H, W]
        self._build_layers(transformerlayers, num_layers)
        self.num_points_cross, self.num_points_self = num_points_cross, num_points_self
        g3 = torch.cat([bev_grid.unsqueeze(2).expand(-1, -1, num_points_cross, -1),
                        torch.linspace(0, Z - 1, num_points_cross).reshape(1, 1, -1, 1).expand(H, W, -1, -1)], -1)
        self.register_buffer('ref_3d', self.mapping.grid2meter(g3).flatten(0, 1).transpose(0, 1).contiguous(), False)
        normed = bev_grid.clone()
        normed[..., 0] = normed[..., 0] / (H - 1)
        normed[..., 1] = normed[..., 1] / (W - 1)
        self.register_buffer('ref_2d', normed, False)

    def _forward_layers_sharded(self, bev_query, key, value, bev_pos, spatial_shapes, level_start_index, cam, mask, ref_2d):
        """Row blocks of the ONE BEV plane per rank (see TPVFormerEncoder._forward_layers_sharded): self-attention of the
        local rows over the full replicated plane, image cross-attention / norms / FFN on the local rows, one all-gather
        per layer; gradients of the unsharded encoder on every rank."""
        from ... import dist as sdist
        from torch.func import functional_call
        if bev_query.shape[0] != 1:
            raise NotImplementedError(f"row_shard=True splits ONE frame over the ranks: batch size must be 1 (got {bev_query.shape[0]})")
        shard = self._row_shard_plan_for([self.bev_size[0] * self.bev_size[1]])
        grad = torch.is_grad_enabled()
        q_full = bev_query
        if grad:
            q_full = sdist.replicate_grad_sum(q_full)
            key = value = sdist.replicate_grad_sum(value)
            bev_pos = sdist.replicate_grad_sum(bev_pos)
        pos_loc, ref_loc = shard.take(bev_pos, 1), shard.take(ref_2d, 1)
        cam_loc, mask_loc = shard.take_plane(cam, 0, 2), shard.take_plane(mask, 0, 2)
        last = len(self.layers) - 1
        for li, layer in enumerate(self.layers):
            call = dict(bev_pos=pos_loc, ref_2d=ref_loc, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                        reference_points_cams=cam_loc, bev_masks=mask_loc, bev_size=self.bev_size, rebatch_plan=None,
                        self_attn_value=q_full)
            q_loc = shard.take(q_full, 1)
            if grad:
                names, ps = zip(*layer.named_parameters())
                out = functional_call(layer, dict(zip(names, sdist.group_grad_sum(ps))), (q_loc, key, value), call)
            else:
                out = layer(q_loc, key, value, **call)
            q_full = sdist.gather_plane_rows(out, shard, reduce_grad=li < last)
        return q_full

    def forward_layers(self, bev_query, key, value, bev_pos=None, spatial_shapes=None, level_start_index=None,
                       img_metas=None, **kwargs):
        bs = bev_query.shape[0]
        cam, mask = point_sampling(self.ref_3d.unsqueeze(0).expand(bs, -1, -1, -1), img_metas)   # read-only
        ref_2d = self.ref_2d.unsqueeze(0).expand(bs, -1, -1, -1).reshape(bs, -1, 1, 2)
        if self._row_sharding():
            return self._forward_layers_sharded(bev_query, key, value, bev_pos, spatial_shapes, level_start_index, cam, mask, ref_2d)
        plan = None   # see TPVFormerEncoder.forward_layers
        for layer in self.layers:
            bev_query = layer(bev_query, key, value, bev_pos=bev_pos, ref_2d=ref_2d, spatial_shapes=spatial_shapes,
                              level_start_index=level_start_index, reference_points_cams=cam, bev_masks=mask,
                              bev_size=self.bev_size, rebatch_plan=plan, **kwargs)
        return bev_query

    def forward(self, representation, ms_img_feats=None, metas=None, **kwargs):
        bs = ms_img_feats[0].shape[0]
        bev_pos, _ = self._positions(bs)
        feat, spatial_shapes, level_start_index = self._flatten_feats(ms_img_feats)
        bev = self.forward_layers(representation, feat, feat, bev_pos=bev_pos, spatial_shapes=spatial_shapes,
                                  level_start_index=level_start_index, img_metas=metas)
        return {'representation': bev}
