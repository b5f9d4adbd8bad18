"""Minimal registry / config shim with the mmcv surface the FCAF3D path uses (mmcv and mmdet are not
installed): `Registry.register_module()`, `build_from_cfg`, the `BACKBONES/HEADS/DETECTORS/LOSSES/
BBOX_ASSIGNERS` registries and `build_backbone/head/loss/detector/assigner`
(mmdet3d/models/builder.py:5-57), and `Config.fromfile` for plain-python configs with `_base_`
inheritance (configs/fcaf3d/*.py)."""
import copy
import os


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, **default_args):
        return build_from_cfg(cfg, self, default_args or None)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
    args = copy.deepcopy(dict(cfg))
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    cls = registry.get(obj_type) if isinstance(obj_type, str) else obj_type
    if cls is None:
        raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    return cls(**args)


BACKBONES = Registry('backbone')
HEADS = Registry('head')
DETECTORS = Registry('detector')
LOSSES = Registry('loss')
BBOX_ASSIGNERS = Registry('bbox_assigner')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_assigner(cfg):
    return BBOX_ASSIGNERS.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return DETECTORS.build(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


def build_model(cfg, train_cfg=None, test_cfg=None):
    return build_detector(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


class ConfigDict(dict):
    """dict with attribute access (mmcv.ConfigDict): `test_cfg.nms_pre`."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(_wrap(v) for v in x)
    return x


def _merge(base, over):
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.pop('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = v
    return out


class Config:
    @staticmethod
    def _load(path):
        ns = {}
        with open(path) as f:
            exec(compile(f.read(), path, 'exec'), ns)
        cfg = {k: v for k, v in ns.items() if (k == '_base_' or not k.startswith('_')) and not callable(v)
               and not isinstance(v, type(os))}          # `_name` = file-local helper value, not a config key
        bases = cfg.pop('_base_', [])
        if isinstance(bases, str):
            bases = [bases]
        merged = {}
        for b in bases:
            merged = _merge(merged, Config._load(os.path.join(os.path.dirname(path), b)))
        return _merge(merged, cfg)

    @staticmethod
    def fromfile(path):
        return _wrap(Config._load(os.path.abspath(path)))
