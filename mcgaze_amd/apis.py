"""``init_detector`` with the reference's semantics (mmdet/apis/inference.py:17-57): build from a
config file or Config, drop train_cfg / pretrained init, load the checkpoint with the key rewrite
``^module\\.`` -> '' (and the leftover ``mask_head`` -> ``blink_head`` rule), set ``model.cfg`` and
``model.CLASSES``, move to the device, ``eval()``.  ``precision`` selects the HIP engine: 'f16x3' (default, parity-grade
fast mode: the reference is fp32 everywhere and this mode reproduces it to < 1e-4 rad), 'fp32' (exact reference mode) or
'f16' / 'bf16' (16-bit throughput modes, fp16 or bf16 storage and MFMA; explicit opt-in: outside the 1e-3 parity tolerance on
random-weight nets, fp16 eight times closer than bf16)."""
import re
import warnings

import torch

from .config import Config
from .registry import build_detector

REVISE_KEYS = [(r'^module\.', ''), ('mask_head', 'blink_head')]
CLASSES = ('face', 'eyes', 'head')


def load_checkpoint(model, filename, map_location='cpu', strict=False, revise_keys=REVISE_KEYS):
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    sd = ckpt.get('state_dict', ckpt)
    for pat, rep in revise_keys:
        sd = {re.sub(pat, rep, k): v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    if missing or unexpected:
        warnings.warn(f'checkpoint {filename}: missing keys {list(missing)[:8]}…, unexpected keys {list(unexpected)[:8]}…')
    return ckpt


def init_detector(config, checkpoint=None, device='cuda:0', cfg_options=None, precision='f16x3'):
    if isinstance(config, str):
        config = Config.fromfile(config)
    elif not isinstance(config, Config):
        raise TypeError(f'config must be a filename or Config object, but got {type(config)}')
    if cfg_options is not None:
        config.merge_from_dict(cfg_options)
    if 'pretrained' in config.model:
        config.model.pretrained = None
    elif 'init_cfg' in config.model.backbone:
        config.model.backbone.init_cfg = None
    config.model.train_cfg = None
    model = build_detector(config.model, test_cfg=config.get('test_cfg'))
    if checkpoint is not None:
        ckpt = load_checkpoint(model, checkpoint, map_location='cpu')
        meta = ckpt.get('meta', {}) if isinstance(ckpt, dict) else {}
        model.CLASSES = meta.get('CLASSES', CLASSES)
    else:
        model.CLASSES = CLASSES
    model.cfg = config
    model.precision = precision
    model.to(device)
    model.eval()
    return model
