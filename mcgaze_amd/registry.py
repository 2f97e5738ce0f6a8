"""Registries with the surface the reference's configs and modules use (mmdet/models/builder.py:7-59,
mmdet/models/utils/builder.py:5-11, mmdet/core/bbox/builder.py:4-21): one ``MODELS`` registry
aliased as BACKBONES / NECKS / ROI_EXTRACTORS / SHARED_HEADS / HEADS / LOSSES / DETECTORS, a
separate ``TRANSFORMER`` registry for DynamicConv and ``BBOX_ASSIGNERS / BBOX_SAMPLERS /
BBOX_CODERS``; ``build_*`` = ``Registry.build(cfg)`` with ``type`` popped and the rest passed as
constructor kwargs."""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f'{obj_cls.__name__}: {e}') from e


class Registry:
    def __init__(self, name, build_func=None, parent=None):
        self._name = name
        self._module_dict = {}
        self.parent = parent
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        return self.parent.get(key) if self.parent is not None else None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f'module must be a class, but got {type(cls)}')
        for n in ([name] if isinstance(name, str) else (name or [cls.__name__])):
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self.name}')
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module
        if inspect.isclass(name):  # @REG.register_module without parentheses
            self._register(name)
            return name

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco


MODELS = Registry('models')
BACKBONES = NECKS = ROI_EXTRACTORS = SHARED_HEADS = HEADS = LOSSES = DETECTORS = MODELS
TRANSFORMER = Registry('Transformer')
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_roi_extractor(cfg):
    return ROI_EXTRACTORS.build(cfg)


def build_shared_head(cfg):
    return SHARED_HEADS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_transformer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:48-59."""
    assert cfg.get('train_cfg') is None or train_cfg is None, 'train_cfg specified in both outer field and model field'
    assert cfg.get('test_cfg') is None or test_cfg is None, 'test_cfg specified in both outer field and model field'
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))
