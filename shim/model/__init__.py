"""PYTHONPATH shim so that the reference's entry scripts run UNEDITED on the MI355X hot path.

    PYTHONPATH=/path/to/this/repo/shim python train.py --py-config config/nuscenes/nuscenes_occ.py ...

`import model` (train.py:71, eval_depth.py:66, eval_iou.py:89, eval_novel_depth.py:66, vis_*.py) then resolves to THIS
package instead of the reference's `model/` (through the meta-path finder shim/sitecustomize.py installs at start-up: a
script's own directory precedes PYTHONPATH, a path entry alone would lose):
  * backbone / neck / segmentor stay the reference's own files (vendor convolutions, the TPVSegmentor control flow):
    the reference's `model/` directory is appended to this package's search path and those three sub-packages are
    imported from it — its `model/__init__.py` (which would also import its lifter / encoder / head) never runs;
  * lifter / encoder / head — every `type=` string of the shipped configs — register from `selfocc_amd.model`.
"""
import importlib
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
if _REPO not in sys.path:
    sys.path.append(_REPO)                      # `selfocc_amd` itself


def _reference_model_dir():
    for p in sys.path:
        d = os.path.join(os.path.abspath(p or os.getcwd()), 'model')
        if os.path.abspath(d) != _HERE and os.path.isfile(os.path.join(d, '__init__.py')) and \
                os.path.isdir(os.path.join(d, 'segmentor')):
            return d
    raise ImportError("selfocc_amd shim: no SelfOcc checkout on sys.path (run the reference's scripts from its root, "
                      "as its README does)")


REFERENCE_MODEL_DIR = _reference_model_dir()
__path__.append(REFERENCE_MODEL_DIR)

import selfocc_amd.model as _ours  # noqa: E402  (lifter / encoder / head under the reference's registry names)

# anything that asks for the reference's hot-path sub-packages by module path gets ours (no double registration)
sys.modules[__name__ + '.lifter'] = importlib.import_module('selfocc_amd.model.lifter')
sys.modules[__name__ + '.encoder'] = importlib.import_module('selfocc_amd.model.encoder')
sys.modules[__name__ + '.head'] = importlib.import_module('selfocc_amd.model.head')
lifter, encoder, head = (sys.modules[__name__ + s] for s in ('.lifter', '.encoder', '.head'))

for _sub in ('backbone', 'neck', 'segmentor'):           # the reference's own (model/__init__.py:1-2, 6)
    _m = importlib.import_module(f'{__name__}.{_sub}')
    globals().update({k: v for k, v in vars(_m).items() if not k.startswith('_')})
