"""Importing this package registers every hot-path class under the reference's names."""
from .bricks import FFN, MultiScaleDeformableAttention  # noqa: F401
from .lifter import TPVQueryLifter, BEVQueryLifter  # noqa: F401
from .encoder import *  # noqa: F401,F403
from .head import NeuSHead  # noqa: F401
