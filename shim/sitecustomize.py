"""Imported automatically at interpreter start-up when this directory is on PYTHONPATH (the `site` module imports
`sitecustomize` if it finds one) — in the launching process AND in every worker `torch.multiprocessing.spawn` starts
(train.py:401), because they inherit the environment.

A script's own directory precedes PYTHONPATH on sys.path, so `import model` / `from loss import OPENOCC_LOSS` inside
train.py (run from the SelfOcc checkout) would find the checkout's packages first.  A meta-path finder runs before the
path search: it hands the two top-level names `model` and `loss` to the packages next to this file (shim/model,
shim/loss), which bind lifter / encoder / head / losses to selfocc_amd and leave backbone / neck / segmentor to the
checkout.  Nothing else is intercepted.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

_SHIM = os.path.dirname(os.path.abspath(__file__))


class _SelfOccHotPathFinder(importlib.abc.MetaPathFinder):
    names = ('model', 'loss')

    def find_spec(self, fullname, path=None, target=None):
        if fullname in self.names and os.environ.get('SELFOCC_SHIM', '1') != '0':
            pkg = os.path.join(_SHIM, fullname)
            return importlib.util.spec_from_file_location(fullname, os.path.join(pkg, '__init__.py'),
                                                          submodule_search_locations=[pkg])
        return None


if not any(isinstance(f, _SelfOccHotPathFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _SelfOccHotPathFinder())

# chain to a sitecustomize this one shadows (distribution / virtualenv hooks), if any
_next = importlib.machinery.PathFinder.find_spec('sitecustomize', [p for p in sys.path if os.path.abspath(p or '.') != _SHIM])
if _next is not None and _next.origin and os.path.abspath(_next.origin) != os.path.abspath(__file__):
    _m = importlib.util.module_from_spec(_next)
    _next.loader.exec_module(_m)
