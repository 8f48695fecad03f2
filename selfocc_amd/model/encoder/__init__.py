from .attention import BEVDeformableAttention, BEVCrossAttention, TPVCrossAttention, CrossViewHybridAttention
from .tpvformer import (TPVPositionalEncoding, BEVPositionalEncoding, TPVFormerLayer, BEVFormerLayer,
                        TPVFormerEncoder, BEVFormerEncoder)
from .utils import point_sampling, get_cross_view_ref_points
