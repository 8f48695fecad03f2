"""Image cross-attention and cross-view hybrid attention of the TPV / BEV lifter — the
MSDA call sites of the reference, behind the same registry names, constructor kwargs,
parameter names (state-dict keys) and forward contracts:

BEVDeformableAttention   <- model/encoder/bevformer/attention/image_cross_attention.py:149-351
BEVCrossAttention        <- model/encoder/bevformer/attention/image_cross_attention.py:12-139
TPVCrossAttention        <- model/encoder/tpvformer/attention/image_cross_attention.py:8-95
CrossViewHybridAttention <- model/encoder/tpvformer/attention/cross_view_hybrid_attention.py:12-124
The sampling itself is the HIP kernel behind MultiScaleDeformableAttnFunction (msda.py).
"""
import math

import os

import torch
import torch.nn as nn

from ...registry import MODELS, build_attention
from ..bricks import (BaseModule, MultiScaleDeformableAttention, TallLinear, constant_init, xavier_init,
                      deformable_sampling, fused_linear)
from ...msda import (msda_cross_inference, MSDACrossFunction, msda_fused_supported, msda_fused_kernels_built,
                     to_head_major)
from .. import bricks


@MODELS.register_module()
class BEVDeformableAttention(BaseModule):
    """Deformable attention with one reference anchor per sampling point
    (num_points == points in the pillar); no output projection / residual — the caller
    (BEVCrossAttention) owns those."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None, value_proj_ratio=1.0):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}')
        self.norm_cfg, self.batch_first = norm_cfg, batch_first
        self.im2col_step, self.embed_dims = im2col_step, embed_dims
        self.num_levels, self.num_heads, self.num_points = num_levels, num_heads, num_points
        self.sampling_offsets = TallLinear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = TallLinear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = TallLinear(embed_dims, int(embed_dims * value_proj_ratio))
        self.init_weights()

    def init_weights(self):
        constant_init(self.sampling_offsets, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        # (the reference deliberately drops mmcv's per-point (i + 1) scaling, :238-239)
        self.sampling_offsets.bias.data = grid_init.view(-1).to(self.sampling_offsets.bias.device)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        self._is_init = True

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        output = deformable_sampling(self, query, value, reference_points, spatial_shapes, level_start_index,
                                     'point', key_padding_mask)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return output


@MODELS.register_module()
class BEVCrossAttention(BaseModule):
    """Every plane query attends to the cameras that see it: visible queries are re-batched
    per camera, sampled by ``deformable_attention``, scattered back and averaged over the
    cameras that hit (image_cross_attention.py:46-139).  The reference's python double loops
    (:101-110, :129-131) are single index_select / index_add_ calls here."""

    def __init__(self, embed_dims=256, num_cams=6, dropout=0.1, init_cfg=None, batch_first=True,
                 deformable_attention=dict(type='MSDeformableAttention3D', embed_dims=256, num_levels=4),
                 **kwargs):
        super().__init__(init_cfg)
        self.dropout = nn.Dropout(dropout)
        self.deformable_attention = build_attention(deformable_attention)
        self.embed_dims, self.num_cams = embed_dims, num_cams
        self.output_proj = TallLinear(embed_dims, embed_dims)
        self.batch_first = batch_first
        self.camera_loop = True     # inference: selfocc_msda_cross_fwd instead of re-batch + scatter-add
        self.init_weight()

    def init_weight(self):
        xavier_init(self.output_proj, distribution='uniform', bias=0.)

    @staticmethod
    def rebatch_plan(bev_masks):
        """Visible (camera, query) pairs and their slots in the per-camera re-batch.  Depends only on
        the camera geometry, so an encoder computes it ONCE per forward and hands it to all its layers
        (``rebatch_plan=`` kwarg): one host sync per plane per frame instead of one per plane per layer.
        Like the reference, visibility is taken from batch element 0 (image_cross_attention.py:92)."""
        vis = bev_masks[:, 0].sum(-1) > 0                                  # (num_cams, Q)
        lens = vis.sum(-1)
        max_len = int(lens.max())                                          # host sync (reference :95)
        cam_idx, q_idx = vis.nonzero(as_tuple=True)                        # sorted by camera, then query
        starts = torch.cumsum(lens, 0) - lens
        slot = torch.arange(cam_idx.numel(), device=bev_masks.device) - starts[cam_idx]
        count = (bev_masks.sum(-1) > 0).permute(1, 2, 0).sum(-1)
        count = torch.clamp(count, min=1.0)
        return cam_idx, q_idx, slot, max_len, count

    def forward(self, query, key, value, residual=None, spatial_shapes=None, reference_points_cams=None,
                bev_masks=None, level_start_index=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if residual is None:
            residual = query
        bs, num_query, _ = query.size()
        num_cams = self.num_cams
        D = reference_points_cams.size(3)
        da = self.deformable_attention
        if (self.camera_loop and bs == 1 and query.is_cuda and isinstance(da, BEVDeformableAttention)
                and da.batch_first and da.num_levels * da.num_points <= 256
                and msda_fused_kernels_built(self.embed_dims // da.num_heads)):
            host = None
            if torch.is_grad_enabled():
                host = getattr(spatial_shapes, '_so_host', None)
                if host is None:
                    host = [int(v) for v in spatial_shapes.reshape(-1).tolist()]
                if not msda_fused_supported(host, num_cams, num_query, da.num_heads, self.embed_dims // da.num_heads,
                                            da.num_levels, da.num_points):
                    host = False
            if host is not False:
                return self._forward_camera_loop(query, value, residual, spatial_shapes, reference_points_cams,
                                                 bev_masks, level_start_index, host, out=kwargs.get('out'),
                                                 value_pre=kwargs.get('value_pre'), post_norm=kwargs.get('post_norm'))
        plan = kwargs.get('rebatch_plan')
        if plan is None:
            plan = self.rebatch_plan(bev_masks)
        cam_idx, q_idx, slot, max_len, count = plan
        queries_rebatch = query.new_zeros([bs, num_cams, max_len, self.embed_dims])
        ref_rebatch = reference_points_cams.new_zeros([bs, num_cams, max_len, D, 2])
        queries_rebatch[:, cam_idx, slot] = query[:, q_idx]
        ref_rebatch[:, cam_idx, slot] = reference_points_cams.permute(1, 0, 2, 3, 4)[:, cam_idx, q_idx]
        queries_rebatch = queries_rebatch.flatten(0, 1)                    # (bs * num_cams, max_len, C)
        ref_rebatch = ref_rebatch.flatten(0, 1)

        _, l, _, _ = key.shape
        key = key.permute(2, 0, 1, 3).reshape(num_cams * bs, l, self.embed_dims)
        value = value.permute(2, 0, 1, 3).reshape(num_cams * bs, l, self.embed_dims)
        sampled = self.deformable_attention(query=queries_rebatch, key=key, value=value,
                                            reference_points=ref_rebatch, spatial_shapes=spatial_shapes,
                                            level_start_index=level_start_index)
        sampled = sampled.view(bs, num_cams, max_len, self.embed_dims)
        slots = torch.zeros_like(query)
        slots.index_add_(1, q_idx, sampled[:, cam_idx, slot])
        slots = slots / count[..., None]
        slots = self.output_proj(slots)
        if bricks.FUSED_DROPOUT_ADD:
            slots = bricks.dropout_add(slots, residual, self.dropout.p, self.dropout.training)
        else:
            slots = self.dropout(slots) + residual
        post_norm = kwargs.get('post_norm')
        return post_norm(slots) if post_norm is not None else slots


    def _forward_camera_loop(self, query, value, residual, spatial_shapes, reference_points_cams, bev_masks,
                             level_start_index, host_shapes=None, out=None, value_pre=None, post_norm=None):
        """No re-batch (inference: plain op; training: MSDACrossFunction under autograd).  The offset / weight linears depend on the query only, so they run once
        on the num_query rows; one HIP launch loops over the cameras that see each query and averages
        (selfocc_msda_cross_fwd) — same arithmetic as the re-batched path, camera order preserved."""
        da = self.deformable_attention
        num_cams, heads, L, P = self.num_cams, da.num_heads, da.num_levels, da.num_points
        _, l, _, _ = value.shape                                            # (cams, nv, bs, C)
        hm = bricks.HEAD_MAJOR_VALUE
        sink = None    # (ValueGradSink, g): where the backward puts grad_value (set by bricks.value_proj_head_major[_multi])
        if value_pre is not None and value_pre.dim() == 4:
            v, hm = value_pre, True    # TPVCrossAttention already laid this plane's values out head-major
            sink = getattr(value_pre, '_so_grad_sink', None)
        elif value_pre is not None:    # this plane's column block of the merged value projection (TPVCrossAttention)
            v, hm = value_pre.view(num_cams, l, heads, -1), False
        else:
            vin = value.permute(2, 0, 1, 3).reshape(num_cams * l, self.embed_dims)
            v_hm = None
            if not hm and da.value_proj.weight.shape[0] == 96:
                # the projection itself writes (cams, heads, l, d) (selfocc_linear_fwd_heads; under autograd _TallLinearHeads)
                v_hm = bricks.value_proj_head_major(da.value_proj.weight, da.value_proj.bias, vin, l, heads)
            if v_hm is not None:
                v, hm = v_hm.view(v_hm.shape[1:]), True      # G = 1: a view, not a select
                s1 = getattr(v_hm, '_so_grad_sink', None)
                sink = (s1, 0) if s1 is not None else None
            else:
                v = da.value_proj(vin.view(num_cams, l, self.embed_dims)).view(num_cams, l, heads, -1)
                if hm:
                    v = to_head_major(v)
        vis_all = getattr(bev_masks, '_so_visible', None)                   # left by the HIP point_sampling
        visible = vis_all[:, 0] if vis_all is not None else bev_masks[:, 0].any(-1)   # (cams, Q), batch element 0 as the reference
        bf16 = bricks.VALUE_BF16 and self.embed_dims // heads == 16     # the bfloat16 gathers exist for d = 16 only
        if bf16 and host_shapes is None and v.dtype != torch.bfloat16:
            v = v.to(torch.bfloat16)
        q2 = query.reshape(-1, query.shape[-1])   # bs == 1; a view, not query[0]: select's backward is a zero fill + a copy
        # sampling_offsets | attention_weights of the (un-rebatched) queries: ONE projection with the stacked weight, read
        # in place by the kernel (bricks.merged_off_logits), else the two Linears
        off = bricks.merged_off_logits(da, q2)
        mlp = (L, P) if off is not None else None
        logits = None
        if off is None:
            off = da.sampling_offsets(q2).view(-1, heads, L, P, 2)
            logits = da.attention_weights(q2).view(-1, heads, L * P)
        if host_shapes is None:
            slots = msda_cross_inference(v, spatial_shapes, level_start_index, reference_points_cams[:, 0], visible,
                                         off, logits, hm, mlp)[None]
        else:
            slots = MSDACrossFunction.apply(v, spatial_shapes, level_start_index, reference_points_cams[:, 0], visible,
                                            off, logits, host_shapes, hm, bf16, mlp, sink)[None]
        if not self.training and not torch.is_grad_enabled():
            # eval: dropout is the identity; output_proj + residual (+ the layer's next norm) in one launch, written
            # straight into the caller's slice of the concatenated plane buffer (`out`)
            return fused_linear(self.output_proj, slots, residual=residual, norm=post_norm, out=out)
        slots = self.output_proj(slots)
        if bricks.FUSED_DROPOUT_ADD:
            slots = bricks.dropout_add(slots, residual, self.dropout.p, self.dropout.training)
        else:
            slots = self.dropout(slots) + residual
        return post_norm(slots) if post_norm is not None else slots


# dev A/B (round 6): SELFOCC_PLANE_STREAMS=1 runs the three planes' inference branches on three streams.  Measured and left OFF:
# the eval encoder gets SLOWER (5.45 -> 5.69 ms at nuscenes_occ, 5.95 -> 6.19 at nuscenes_depth, same box, twice; results unchanged,
# scripts/diag/plane_streams_ab.sh) — the gather kernels already keep the texture path 70 - 77 % busy on their own and lose more
# to each other than the short launches gain by hiding under them.
PLANE_STREAMS = os.environ.get('SELFOCC_PLANE_STREAMS', '0') == '1'
_PLANE_STREAMS = {}


def _plane_streams(device):
    key = str(device)
    if key not in _PLANE_STREAMS:
        _PLANE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(3)]
    return _PLANE_STREAMS[key]


@MODELS.register_module()
class TPVCrossAttention(BaseModule):
    """One BEVCrossAttention per TPV plane (hw, zh, wz) with num_points = [wz, zh, hw] pillar
    sizes (tpvformer/attention/image_cross_attention.py:19-69)."""

    def __init__(self, embed_dims=256, num_cams=6, dropout=0.1, init_cfg=None, batch_first=True,
                 num_heads=16, num_levels=4, num_points=[64, 64, 8]):
        super().__init__(init_cfg)

        def plane(points):
            return build_attention(dict(
                type='BEVCrossAttention', embed_dims=embed_dims, num_cams=num_cams, dropout=dropout,
                batch_first=batch_first,
                deformable_attention=dict(type='BEVDeformableAttention', embed_dims=embed_dims,
                                          num_heads=num_heads, num_levels=num_levels, num_points=points,
                                          dropout=dropout, batch_first=batch_first)))
        self.attn_hw = plane(num_points[2])
        self.attn_zh = plane(num_points[1])
        self.attn_wz = plane(num_points[0])
        self.attns = [self.attn_hw, self.attn_zh, self.attn_wz]
        self.embed_dims = embed_dims

    def forward(self, query, key, value, residual=None, spatial_shapes=None, reference_points_cams=None,
                tpv_masks=None, level_start_index=None, **kwargs):
        plans = kwargs.get('rebatch_plans') or [None] * 3
        outs, vpre = [None] * 3, [None] * 3
        if not torch.is_grad_enabled() and not self.training and query[0].shape[0] == 1:
            # inference: the three planes' results land in consecutive slices of ONE buffer, so that the layer's next
            # norm / ffn step sees the concatenated tensor without a copy (tpvformer.cat_planes)
            sizes = [q.shape[1] for q in query]
            buf = query[0].new_empty(1, sum(sizes), query[0].shape[-1])
            outs = list(torch.split(buf, sizes, 1))
            if value.is_cuda and all(a.camera_loop for a in self.attns):
                # ... and the three planes' value projections of the SAME image features are one GEMM with N = 3 C
                # (the 68 MB input is read once instead of three times); each plane's kernel reads its column block
                # in place (selfocc_msda_cross_fwd value_stride)
                C = self.embed_dims
                w, b = self._merged_value_proj()
                cams, l = value.shape[0], value.shape[1]
                vin = value.permute(2, 0, 1, 3).reshape(cams * l, C)
                heads0 = self.attns[0].deformable_attention.num_heads
                v_hm = bricks.value_proj_head_major(w, b, vin, l, heads0) if C == 96 else None
                if v_hm is not None:
                    # (3, cams, heads, l, d): each plane's head-major value, written by the projection kernel itself
                    if bricks.VALUE_BF16 and C // heads0 == 16:
                        v_hm = v_hm.to(torch.bfloat16)
                    vpre = list(v_hm)

                    def plane(i):
                        return self.attns[i](query[i], key, value, residual[i] if residual is not None else None,
                                             spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                             reference_points_cams=reference_points_cams[i], bev_masks=tpv_masks[i],
                                             rebatch_plan=plans[i], out=outs[i], value_pre=vpre[i],
                                             post_norm=kwargs.get('post_norm'))
                    if not PLANE_STREAMS:
                        return [plane(i) for i in range(3)]
                    # the three planes are independent until the layer's next step: each runs on its own stream (its
                    # temporaries live and die in that stream's pool; inputs and the `outs` slices belong to the caller's
                    # stream, which waits for all three before it goes on)
                    main = torch.cuda.current_stream()
                    side = _plane_streams(value.device)
                    start = torch.cuda.Event()
                    start.record(main)
                    res = []
                    for i in range(3):
                        side[i].wait_event(start)
                        with torch.cuda.stream(side[i]):
                            res.append(plane(i))
                    for st in side:
                        main.wait_stream(st)
                    return res
                if bricks.FUSED_LINEAR_FWD and bricks._linear_fwd_ok(vin, w):
                    v_all = bricks.linear_fwd(vin, w, b).view(cams, l, 3 * C)
                else:
                    v_all = torch.addmm(b, vin, w.t()).view(cams, l, 3 * C)
                if bricks.VALUE_BF16 and C // heads0 == 16:
                    v_all = v_all.to(torch.bfloat16)       # one cast for the three planes
                if bricks.HEAD_MAJOR_VALUE:
                    # one transposing copy for the three planes: (plane, cams, heads, l, d), each plane dense
                    heads = self.attns[0].deformable_attention.num_heads
                    vpre = list(v_all.view(cams, l, 3, heads, C // heads).permute(2, 0, 3, 1, 4).contiguous())
                else:
                    vpre = [v_all[..., i * C:(i + 1) * C] for i in range(3)]
        elif (torch.is_grad_enabled() and value.is_cuda and query[0].shape[0] == 1 and value.shape[2] == 1
              and all(a.camera_loop for a in self.attns)):
            # training: the three planes' value projections of the same image features are one autograd node
            # (bricks._TallLinearHeadsMulti): one projection, one weight / input gradient pass per layer
            cams, l = value.shape[0], value.shape[1]
            das = [a.deformable_attention for a in self.attns]
            v3 = bricks.value_proj_head_major_multi([da.value_proj for da in das],
                                                    value.permute(2, 0, 1, 3).reshape(cams * l, self.embed_dims), l,
                                                    das[0].num_heads) if len({da.num_heads for da in das}) == 1 else None
            if v3 is not None:
                vpre = v3          # (cams, heads, l, d) each: `_forward_camera_loop` takes a 4-d value_pre as head-major
        return [self.attns[i](query[i], key, value, residual[i] if residual is not None else None,
                              spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                              reference_points_cams=reference_points_cams[i], bev_masks=tpv_masks[i],
                              rebatch_plan=plans[i], out=outs[i], value_pre=vpre[i], post_norm=kwargs.get('post_norm'))
                for i in range(3)]

    def _merged_value_proj(self):
        """(3 C, C) weight and (3 C) bias of the three planes' value_proj stacked; cached until a parameter changes."""
        ps = [a.deformable_attention.value_proj for a in self.attns]
        key = tuple((p.weight._version, p.weight.data_ptr(), p.bias._version, p.bias.data_ptr()) for p in ps)
        if getattr(self, '_vp_cache', (None,))[0] != key:
            self._vp_cache = (key, torch.cat([p.weight for p in ps], 0).detach(), torch.cat([p.bias for p in ps], 0).detach())
        return self._vp_cache[1], self._vp_cache[2]


@MODELS.register_module()
class CrossViewHybridAttention(MultiScaleDeformableAttention):
    """Self-attention across the three TPV planes treated as 3 "levels"; reference points
    carry their own (level, point) dims (cross_view_hybrid_attention.py:96-99)."""
    _reference_kind = 'level_point'
