"""Learned query lifters (config surface only — no image math):
TPVQueryLifter <- model/lifter/tpv_query_lifter.py:7-36, BEVQueryLifter <- bev_query_lifter.py:7-26."""
import torch
import torch.nn as nn

from ..registry import MODELS
from .bricks import BaseModule


@MODELS.register_module()
class TPVQueryLifter(BaseModule):
    def __init__(self, tpv_h, tpv_w, tpv_z, dim, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.tpv_h, self.tpv_w, self.tpv_z, self.dim = tpv_h, tpv_w, tpv_z, dim
        self.tpv_hw = nn.Parameter(torch.randn(1, tpv_h * tpv_w, dim))
        self.tpv_zh = nn.Parameter(torch.randn(1, tpv_z * tpv_h, dim))
        self.tpv_wz = nn.Parameter(torch.randn(1, tpv_w * tpv_z, dim))

    def forward(self, ms_img_feats, *args, **kwargs):
        bs = ms_img_feats[0].shape[0]
        params = (self.tpv_hw, self.tpv_zh, self.tpv_wz)
        if bs == 1 and not torch.is_grad_enabled() and params[0].is_cuda:
            # inference: the planes as views of ONE concatenated tensor (kept until a parameter changes), the form the
            # encoder's first step wants (tpvformer._Planes): no per-frame batch copies, no concatenating copy
            from .encoder.tpvformer import _as_planes
            key = tuple((None if p.is_inference() else p._version, p.data_ptr()) for p in params)
            hit = getattr(self, '_cat_cache', None)
            if hit is None or hit[0] != key:
                hit = self._cat_cache = (key, torch.cat([p.detach() for p in params], 1))
            return {'representation': _as_planes(hit[1], [p.shape[1] for p in params])}
        # read-only downstream: an expanded view instead of the reference's .repeat (a copy, and a reduction in its backward)
        return {'representation': [p.expand(bs, -1, -1) for p in params]}


@MODELS.register_module()
class BEVQueryLifter(BaseModule):
    def __init__(self, bev_h, bev_w, dim, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.bev_h, self.bev_w, self.dim = bev_h, bev_w, dim
        self.bev = nn.Parameter(torch.randn(1, bev_h * bev_w, dim))

    def forward(self, ms_img_feats, *args, **kwargs):
        bs = ms_img_feats[0].shape[0]
        return {'representation': self.bev.to(ms_img_feats[0].dtype).expand(bs, -1, -1)}
