"""PYTHONPATH shim: `from loss import OPENOCC_LOSS` (train.py:73) resolves to the MI355X loss layer — same registry
name ('openocc_loss'), class names, constructor kwargs and `input_dict` remapping as loss/__init__.py:1-11."""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.append(_REPO)

from selfocc_amd.loss import *  # noqa: E402,F401,F403
from selfocc_amd.loss import OPENOCC_LOSS  # noqa: E402,F401
