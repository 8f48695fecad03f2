"""Minimal ``Config.fromfile`` for mmengine-style python configs (``_base_`` inheritance,
recursive dict merge, ``_delete_``) so that the reference's shipped configs
(config/**/*.py) parse UNCHANGED when mmengine is absent.  With mmengine installed use
``mmengine.Config`` — the resulting dicts are interchangeable."""
import copy
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(_wrap(v) for v in x)
    return x


def _merge(base, new):
    out = copy.deepcopy(base)
    for k, v in new.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = copy.deepcopy(v)
    return out


class Config:
    def __init__(self, cfg_dict, filename=None):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def _load(path):
        path = os.path.abspath(path)
        ns = {'__file__': path}
        with open(path) as f:
            exec(compile(f.read(), path, 'exec'), ns)  # configs are python, as in mmengine
        cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v)
               and not isinstance(v, type(os))}
        bases = cfg.pop('_base_', [])
        if isinstance(bases, str):
            bases = [bases]
        merged = {}
        for b in bases:
            merged = _merge(merged, Config._load(os.path.join(os.path.dirname(path), b)))
        return _merge(merged, cfg)

    @classmethod
    def fromfile(cls, path):
        return cls(cls._load(path), filename=path)

    def __getattr__(self, k):
        return getattr(self._cfg_dict, k)

    def __getitem__(self, k):
        return self._cfg_dict[k]

    def __contains__(self, k):
        return k in self._cfg_dict

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))
