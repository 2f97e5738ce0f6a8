"""DEV-ONLY (build container): permissive stand-in for the absent ``mmcv`` package so that
``/root/reference/mmdet`` can be imported on CPU to validate ``oracle/mcgaze_oracle.py`` and
to generate ``tests/golden/*.npz`` (oracle/dev/make_goldens.py).

Not shipped as "the reference", never imported by the product or by any test that runs on
the GPU box (``/root/reference`` does not exist there).  It fabricates any ``mmcv.*`` /
``cv2`` / ``pycocotools`` / ... module with permissive dummies, plus real implementations of
the few mmcv pieces the hot path executes (Registry, BaseModule, build_norm_layer,
ConvModule, MultiheadAttention, FFN, RoIAlign) -- thin wrappers over torch whose semantics
follow mmcv-full 1.4.8 as published.
"""
import sys, types, importlib.abc, importlib.machinery, inspect, functools, copy, math
import torch, torch.nn as nn, torch.nn.functional as F

PREFIXES = ('mmcv', 'pycocotools', 'terminaltables', 'cv2', 'torchvision', 'cityscapesscripts', 'lvis', 'panopticapi', 'imagecorruptions', 'albumentations', 'onnx', 'onnxruntime', 'timm', 'sklearn_unused')

class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _mk(name)
class _Dummy(metaclass=_Meta):
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Dummy()
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Dummy()
    def __iter__(self): return iter(())
def _mk(name):
    return _Meta(name, (_Dummy,), {})

class _Mod(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        v = _mk(name)
        setattr(self, name, v)
        return v

class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
    def create_module(self, spec):
        m = _Mod(spec.name); m.__path__ = []; return m
    def exec_module(self, module): pass
sys.meta_path.insert(0, _Finder())

import mmcv
mmcv.__version__ = '1.4.8'

# ---------------- Registry ----------------
def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items(): args.setdefault(k, v)
    t = args.pop('type')
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None: raise KeyError(f'{t} not in {registry.name}')
    return cls(**args)
class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self.name = name; self._d = {}; self.parent = parent
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)
    def get(self, k):
        if k in self._d: return self._d[k]
        if self.parent is not None: return self.parent.get(k)
        return None
    def __contains__(self, k): return self.get(k) is not None
    @property
    def module_dict(self): return self._d
    def build(self, *a, **k): return self.build_func(*a, **k, registry=self)
    def _reg(self, cls, name=None, force=False):
        for n in ([name] if isinstance(name, str) else (name or [cls.__name__])):
            self._d[n] = cls
    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._reg(module, name, force); return module
        if isinstance(name, type):  # used as @x.register_module without ()
            self._reg(name); return name
        def deco(cls): self._reg(cls, name, force); return cls
        return deco
import mmcv.utils as mu
mu.Registry = Registry; mu.build_from_cfg = build_from_cfg
mu.TORCH_VERSION = torch.__version__
def digit_version(v): return tuple(int(x) for x in v.split('+')[0].split('.')[:3] if x.isdigit())
mu.digit_version = digit_version
mu.is_str = lambda x: isinstance(x, str)
mu.is_tuple_of = lambda s, t: isinstance(s, tuple) and all(isinstance(i, t) for i in s)
mu.is_list_of = lambda s, t: isinstance(s, list) and all(isinstance(i, t) for i in s)
mu.is_seq_of = lambda s, t, seq_type=None: all(isinstance(i, t) for i in s)
mu.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
mmcv.is_str = mu.is_str; mmcv.is_tuple_of = mu.is_tuple_of; mmcv.is_list_of = mu.is_list_of; mmcv.is_seq_of = mu.is_seq_of
mmcv.jit = lambda *a, **k: (lambda f: f)
class ConfigDict(dict):
    def __getattr__(self, k):
        try: return self[k]
        except KeyError: raise AttributeError(k)
    def __setattr__(self, k, v): self[k] = v
mmcv.ConfigDict = ConfigDict; mu.ConfigDict = ConfigDict

# ---------------- runner ----------------
import mmcv.runner as mr
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__(); self._is_init = False; self.init_cfg = copy.deepcopy(init_cfg)
    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'): m.init_weights()
class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg); nn.ModuleList.__init__(self, modules)
class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg); nn.Sequential.__init__(self, *args)
mr.BaseModule = BaseModule; mr.ModuleList = ModuleList; mr.Sequential = Sequential
def _noop_deco(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k: return a[0]
    return lambda f: f
mr.auto_fp16 = _noop_deco; mr.force_fp32 = _noop_deco
mr.get_dist_info = lambda: (0, 1)
import mmcv.runner.base_module as mrb
mrb.BaseModule = BaseModule; mrb.ModuleList = ModuleList; mrb.Sequential = Sequential

# ---------------- cnn ----------------
import mmcv.cnn as mc
mc.MODELS = Registry('model')
def build_conv_layer(cfg, *a, **k):
    assert cfg is None or cfg.get('type') in ('Conv2d', 'Conv'), cfg
    return nn.Conv2d(*a, **k)
def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg); t = cfg.pop('type'); rg = cfg.pop('requires_grad', True); cfg.setdefault('eps', 1e-5)
    if t == 'BN': layer, ab = nn.BatchNorm2d(num_features, **cfg), 'bn'
    elif t == 'LN': layer, ab = nn.LayerNorm(num_features, **cfg), 'ln'
    elif t == 'GN': layer, ab = nn.GroupNorm(num_channels=num_features, **cfg), 'gn'
    else: raise KeyError(t)
    for p in layer.parameters(): p.requires_grad = rg
    return ab + str(postfix), layer
def build_activation_layer(cfg):
    cfg = dict(cfg); t = cfg.pop('type')
    return {'ReLU': nn.ReLU, 'GELU': nn.GELU, 'Sigmoid': nn.Sigmoid}[t](**cfg)
class ConvModule(nn.Module):
    def __init__(self, in_c, out_c, k, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **kw):
        super().__init__()
        self.with_norm = norm_cfg is not None; self.with_activation = act_cfg is not None
        if bias == 'auto': bias = not self.with_norm
        self.conv = nn.Conv2d(in_c, out_c, k, stride, padding, dilation, groups, bias)
        if self.with_norm:
            self.norm_name, n = build_norm_layer(norm_cfg, out_c); self.add_module(self.norm_name, n)
        if self.with_activation:
            a = dict(act_cfg); a.setdefault('inplace', inplace); self.activate = build_activation_layer(a)
    def forward(self, x):
        x = self.conv(x)
        if self.with_norm: x = getattr(self, self.norm_name)(x)
        if self.with_activation: x = self.activate(x)
        return x
def bias_init_with_prob(p): return float(-math.log((1 - p) / p))
mc.build_conv_layer = build_conv_layer; mc.build_norm_layer = build_norm_layer
mc.build_activation_layer = build_activation_layer; mc.ConvModule = ConvModule
mc.bias_init_with_prob = bias_init_with_prob
mc.build_plugin_layer = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
import mmcv.cnn.bricks.transformer as mt
class MultiheadAttention(BaseModule):
    """mmcv 1.4.x semantics as recalled: identity + proj_drop(nn.MultiheadAttention(q,k,v)[0]); seq-first."""
    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=dict(type='Dropout', drop_prob=0.), init_cfg=None, batch_first=False, **kw):
        super().__init__(init_cfg)
        self.embed_dims = embed_dims; self.num_heads = num_heads; self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kw)
        self.proj_drop = nn.Dropout(proj_drop); self.dropout_layer = nn.Identity()
    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None, key_padding_mask=None, **kw):
        if key is None: key = query
        if value is None: value = key
        if identity is None: identity = query
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask, key_padding_mask=key_padding_mask)[0]
        return identity + self.dropout_layer(self.proj_drop(out))
class FFN(BaseModule):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None, add_identity=True, init_cfg=None, **kw):
        super().__init__(init_cfg)
        if 'dropout' in kw: ffn_drop = kw['dropout']
        layers = []; c = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(nn.Linear(c, feedforward_channels), build_activation_layer(act_cfg), nn.Dropout(ffn_drop))); c = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims)); layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers); self.dropout_layer = nn.Identity(); self.add_identity = add_identity
    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity: return self.dropout_layer(out)
        if identity is None: identity = x
        return identity + self.dropout_layer(out)
mt.MultiheadAttention = MultiheadAttention; mt.FFN = FFN
import mmcv.cnn.bricks.registry as mreg
for n in ('ATTENTION', 'FEEDFORWARD_NETWORK', 'TRANSFORMER_LAYER', 'TRANSFORMER_LAYER_SEQUENCE', 'DROPOUT_LAYERS', 'NORM_LAYERS', 'CONV_LAYERS', 'PLUGIN_LAYERS', 'ACTIVATION_LAYERS', 'POSITIONAL_ENCODING', 'UPSAMPLE_LAYERS', 'PADDING_LAYERS'):
    setattr(mreg, n, Registry(n))
mt.TRANSFORMER_LAYER = mreg.TRANSFORMER_LAYER; mt.TRANSFORMER_LAYER_SEQUENCE = mreg.TRANSFORMER_LAYER_SEQUENCE
mt.BaseTransformerLayer = type('BaseTransformerLayer', (BaseModule,), {})
mt.TransformerLayerSequence = type('TransformerLayerSequence', (BaseModule,), {})
mt.build_transformer_layer_sequence = lambda *a, **k: None

# ---------------- ops.RoIAlign (pure torch, aligned=True, avg) ----------------
import mmcv.ops as mo
class RoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True, use_torchvision=False):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = float(spatial_scale); self.sampling_ratio = int(sampling_ratio); self.aligned = aligned
        assert pool_mode == 'avg' and sampling_ratio > 0
    def forward(self, feat, rois):
        N, C, H, W = feat.shape; K = rois.shape[0]; ph, pw = self.output_size; s = self.sampling_ratio
        off = 0.5 if self.aligned else 0.0
        b = rois[:, 0].long()
        x1 = rois[:, 1] * self.spatial_scale - off; y1 = rois[:, 2] * self.spatial_scale - off
        x2 = rois[:, 3] * self.spatial_scale - off; y2 = rois[:, 4] * self.spatial_scale - off
        rw = x2 - x1; rh = y2 - y1
        if not self.aligned: rw = rw.clamp(min=1.); rh = rh.clamp(min=1.)
        bw = rw / pw; bh = rh / ph
        iy = (torch.arange(ph * s, dtype=feat.dtype) + 0.5) / s   # in units of bins
        ix = (torch.arange(pw * s, dtype=feat.dtype) + 0.5) / s
        ys = y1[:, None] + iy[None] * bh[:, None]   # K, ph*s
        xs = x1[:, None] + ix[None] * bw[:, None]
        def prep(c, L):
            invalid = (c < -1.0) | (c > L)
            c = c.clamp(min=0)
            lo = c.floor().long(); hi = lo + 1
            top = lo >= L - 1
            lo = torch.where(top, torch.full_like(lo, L - 1), lo); hi = torch.where(top, torch.full_like(hi, L - 1), hi)
            c = torch.where(top, lo.to(c.dtype), c)
            l = c - lo.to(c.dtype); h = 1 - l
            return lo, hi, l, h, invalid
        ylo, yhi, ly, hy, yinv = prep(ys, H); xlo, xhi, lx, hx, xinv = prep(xs, W)
        fb = feat[b]  # K,C,H,W
        def g(yi, xi):
            idx = (yi[:, :, None] * W + xi[:, None, :]).reshape(K, 1, -1).expand(K, C, -1)
            return fb.reshape(K, C, H * W).gather(2, idx).reshape(K, C, ph * s, pw * s)
        v = (g(ylo, xlo) * (hy[:, :, None] * hx[:, None, :])[:, None] + g(ylo, xhi) * (hy[:, :, None] * lx[:, None, :])[:, None]
             + g(yhi, xlo) * (ly[:, :, None] * hx[:, None, :])[:, None] + g(yhi, xhi) * (ly[:, :, None] * lx[:, None, :])[:, None])
        inv = (yinv[:, :, None] | xinv[:, None, :])[:, None]
        v = torch.where(inv, torch.zeros_like(v), v)
        return v.reshape(K, C, ph, s, pw, s).mean(dim=(3, 5))
mo.RoIAlign = RoIAlign
