"""Loss layer behind the reference's OPENOCC_LOSS registry (loss/__init__.py:1-11): same
class names, constructor kwargs, ``input_dict`` key remapping (loss/base_loss.py:34-39) and
return values.  The per-sample reprojection sampling runs in csrc/reproj.hip."""
from ..registry import OPENOCC_LOSS
from .base import BaseLoss, MultiLoss
from .reproj import ReprojLossMonoMultiNewCombine, ReprojLossMonoMultiNew, SSIM
from .simple import (RGBLossMS, SemLossMS, SemCELossMS, EikonalLoss, SecondGradLoss, EdgeLoss3DMS,
                     SparsityLoss, HardSparsityLoss, SoftSparsityLoss, AdaptiveSparsityLoss)

__all__ = ['OPENOCC_LOSS', 'BaseLoss', 'MultiLoss', 'ReprojLossMonoMultiNewCombine', 'ReprojLossMonoMultiNew',
           'SSIM', 'RGBLossMS', 'SemLossMS', 'SemCELossMS', 'EikonalLoss', 'SecondGradLoss', 'EdgeLoss3DMS',
           'SparsityLoss', 'HardSparsityLoss', 'SoftSparsityLoss', 'AdaptiveSparsityLoss']
