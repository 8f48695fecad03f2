"""Registry / builder surface of the drop-in boundary.

The reference selects every hot-path class by ``type=`` strings in mmengine configs
(mmseg ``MODELS/HEADS/SEGMENTORS``, ``mmcv.cnn.bricks.transformer.build_attention`` ...,
``OPENOCC_LOSS`` in loss/__init__.py:2).  When mmengine is importable our classes register
into ITS registries (true drop-in); otherwise — mmengine / mmcv / mmseg are absent in this
image — an API-compatible minimal ``Registry`` is used.
"""
import inspect

try:  # pragma: no cover - not installed in the build image
    from mmengine.registry import Registry, MODELS as _MM_MODELS
    HAVE_MMENGINE = True
except Exception:  # noqa: BLE001
    HAVE_MMENGINE = False

    class Registry:
        def __init__(self, name, parent=None, **_):
            self.name = name
            self._module_dict = {}
            self.parent = parent

        def register_module(self, name=None, force=False, module=None):
            def _reg(cls):
                key = name or cls.__name__
                if key in self._module_dict and not force and self._module_dict[key] is not cls:
                    raise KeyError(f"{key} is already registered in {self.name}")
                self._module_dict[key] = cls
                return cls
            if module is not None:
                return _reg(module)
            return _reg

        def get(self, key):
            if key in self._module_dict:
                return self._module_dict[key]
            if self.parent is not None:
                return self.parent.get(key)
            return None

        def build(self, cfg, **default_args):
            if cfg is None:
                return None
            if not isinstance(cfg, dict) or 'type' not in cfg:
                raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
            args = dict(cfg)
            for k, v in default_args.items():
                args.setdefault(k, v)
            typ = args.pop('type')
            cls = self.get(typ) if isinstance(typ, str) else typ
            if cls is None:
                raise KeyError(f"{typ} is not in the {self.name} registry")
            return cls(**args)

        def __contains__(self, key):
            return self.get(key) is not None


if HAVE_MMENGINE:  # pragma: no cover
    MODELS = _MM_MODELS
else:
    MODELS = Registry('model')
# mmseg's HEADS / SEGMENTORS are aliases of MODELS in mmseg >= 1.0 (mmseg/registry)
HEADS = SEGMENTORS = ATTENTION = FEEDFORWARD_NETWORK = POSITIONAL_ENCODING = TRANSFORMER_LAYER = MODELS
OPENOCC_LOSS = Registry('openocc_loss')


def build_from_cfg(cfg, registry=MODELS, **default_args):
    return registry.build(cfg, **default_args)


# mmcv.cnn.bricks.transformer builders used by the reference
build_attention = build_feedforward_network = build_positional_encoding = build_transformer_layer = build_from_cfg
build_head = build_segmentor = build_from_cfg
