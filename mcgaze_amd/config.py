"""Python-file configs with the semantics the reference's configs rely on (``mmcv.Config.fromfile``,
SURVEY.md section 8(b)): the file is executed as Python (module-level variables, list
comprehensions, ``**splat`` all work), ``_base_`` (str or list) is loaded first and merged
dict-by-dict, a child dict carrying ``_delete_=True`` replaces the inherited one instead of
merging into it, lists are replaced wholesale, and dotted ``--cfg-options`` keys can be merged in
afterwards.  ``configs/multiclue_gaze/*.py`` of the reference load unchanged (tests/test_config.py).
"""
import copy
import os
import types

DELETE_KEY = '_delete_'
BASE_KEY = '_base_'


class ConfigDict(dict):
    """dict with attribute access; missing attribute -> AttributeError (so hasattr works)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'") from None

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


def _merge(child, base):
    """Merge ``child`` into a copy of ``base`` (child wins)."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get(DELETE_KEY, False):
            out[k] = _merge(v, out[k])
        elif isinstance(v, dict):
            v = dict(v)
            v.pop(DELETE_KEY, None)
            out[k] = _strip_delete(v)
        else:
            out[k] = v
    return out


def _strip_delete(d):
    if isinstance(d, dict):
        return {k: _strip_delete(v) for k, v in d.items() if k != DELETE_KEY}
    if isinstance(d, list):
        return [_strip_delete(v) for v in d]
    return d


def _exec_file(path):
    path = os.path.abspath(os.path.expanduser(path))
    if not os.path.isfile(path):
        raise FileNotFoundError(f'config file not found: {path}')
    if not path.endswith('.py'):
        raise IOError('only .py configs are supported')
    ns = {'__file__': path, '__name__': '__mcgaze_config__'}
    with open(path, 'r', encoding='utf-8') as f:
        exec(compile(f.read(), path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = cfg.pop(BASE_KEY, None)
    if bases is not None:
        merged = {}
        for b in ([bases] if isinstance(bases, str) else list(bases)):
            bcfg = _exec_file(os.path.join(os.path.dirname(path), b))
            dup = set(merged) & set(bcfg)
            if dup:
                raise KeyError(f'duplicate keys in base configs of {path}: {sorted(dup)}')
            merged.update(bcfg)
        cfg = _merge(cfg, merged)
    return _strip_delete(cfg)


class Config:
    """``Config.fromfile(path)`` -> attribute-style access to the merged config."""

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_exec_file(filename), filename=os.path.abspath(filename))

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def items(self):
        return self._cfg_dict.items()

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options):
        """``{'model.roi_head.num_stages': 2}``-style overrides (tools/test_gaze360_gaze.py:33-42)."""
        nested = {}
        for full_key, v in options.items():
            d = nested
            parts = full_key.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge(nested, self.to_dict())))

    def __repr__(self):
        return f'Config (path: {self._filename}): {dict(self._cfg_dict)!r}'
