"""The drop-in boundary on CPU (-m "not gpu"): config loader, registry, module surface with the
reference's state_dict layout, loud failure without a GPU, and that libmcgaze_hip.so loads and
exports every symbol include/mcgaze_hip.h declares (no compute calls here)."""
import os
import re

import numpy as np
import pytest
import torch

import mcgaze_amd
from mcgaze_amd import Config, build_detector, synth
from mcgaze_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OWN_CFG = os.path.join(ROOT, 'configs', 'mcgaze', 'r50_clip7_gaze360.py')
REF_CFGS = ['/root/reference/configs/multiclue_gaze/multiclue_gaze_r50_gaze360.py',
            '/root/reference/configs/multiclue_gaze/multiclue_gaze_r50_l2cs.py']


def build(cfg_path):
    cfg = Config.fromfile(cfg_path)
    cfg.model.train_cfg = None
    if 'init_cfg' in cfg.model.backbone:
        cfg.model.backbone.init_cfg = None
    return cfg, build_detector(cfg.model)


def test_own_config_builds_reference_state_dict_layout():
    cfg, model = build(OWN_CFG)
    assert cfg.clip_length == 7 and cfg.data.test.pipeline[1].type == 'CenterCrop' and cfg.dist_params.backend == 'nccl'
    ref = synth.make_state_dict(0)
    sd = model.state_dict()
    assert set(sd) == set(ref) and len(sd) == 744
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(np.asarray(ref[k]).shape), k
    assert sum(v.numel() for v in sd.values()) == 83367453  # SURVEY.md section 0 [probe]
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}, strict=True)


@pytest.mark.skipif(not os.path.exists(REF_CFGS[0]), reason='reference tree only exists in the build container')
@pytest.mark.parametrize('path', REF_CFGS)
def test_reference_configs_load_unchanged(path):
    cfg = Config.fromfile(path)
    # _base_ chain + _delete_ semantics (multiclue_gaze_r50_gaze360.py:100-112)
    assert cfg.optimizer.type == 'AdamW' and 'momentum' not in cfg.optimizer
    assert cfg.runner == dict(type='IterBasedRunner', max_iters=7000 if 'gaze360.py' in path else 13000)
    assert cfg.lr_config.warmup == 'linear' and cfg.lr_config.warmup_iters == 1000  # merged, not replaced
    assert len(cfg.model.roi_head.bbox_head) == 4 and cfg.model.roi_head.bbox_head[0].bbox_coder.target_stds == [0.5, 0.5, 1., 1.]
    if 'l2cs' in path:
        assert cfg.data.test.pipeline[1].img_scale == (448, 448) and cfg.data.samples_per_gpu == 8
    for with_train_cfg in (True, False):  # train_cfg names assigner/sampler types that must at least parse
        c = Config.fromfile(path)
        c.model.backbone.init_cfg = None
        if not with_train_cfg:
            c.model.train_cfg = None
        model = build_detector(c.model)
        assert set(model.state_dict()) == set(synth.make_state_dict(0))


def test_cfg_options_merge():
    cfg = Config.fromfile(OWN_CFG)
    cfg.merge_from_dict({'model.roi_head.num_stages': 4, 'data.samples_per_gpu': 8, 'new.key': 1})
    assert cfg.data.samples_per_gpu == 8 and cfg.new.key == 1 and cfg.model.neck.out_channels == 256


def test_registry_semantics():
    from mcgaze_amd.registry import Registry, build_from_cfg
    reg = Registry('toy')

    @reg.register_module()
    class A:
        def __init__(self, x, y=2):
            self.x, self.y = x, y
    assert reg.build(dict(type='A', x=1)).y == 2 and build_from_cfg(dict(type=A, x=3), reg, dict(y=5)).y == 5
    with pytest.raises(KeyError):
        reg.build(dict(type='Missing'))
    with pytest.raises(KeyError):
        reg.register_module(module=A)
    assert mcgaze_amd.MODELS.get('MultiClueGaze') is not None and mcgaze_amd.TRANSFORMER.get('DynamicConv') is not None
    assert mcgaze_amd.BACKBONES is mcgaze_amd.HEADS is mcgaze_amd.DETECTORS


def test_unsupported_options_fail_loudly():
    cfg = Config.fromfile(OWN_CFG)
    cfg.model.backbone.style = 'caffe'
    with pytest.raises(NotImplementedError):
        build_detector(cfg.model)
    cfg = Config.fromfile(OWN_CFG)
    cfg.model.backbone.depth = 18
    with pytest.raises(KeyError):
        build_detector(cfg.model)


def test_forward_argument_errors_and_no_cpu_fallback():
    _, model = build(OWN_CFG)
    img = torch.zeros(7, 3, 224, 224)
    metas = synth.make_img_metas(7)
    with pytest.raises(TypeError):          # base.py:122-124
        model(img=img, img_metas=[metas], return_loss=False)
    with pytest.raises(ValueError):         # base.py:127-129
        model(img=[img, img], img_metas=[metas], return_loss=False)
    with pytest.raises(NotImplementedError):
        model(img=[img], img_metas=[metas], return_loss=True)
    if not torch.cuda.is_available():
        with pytest.raises(L.McgError):     # the product path refuses to run without the HIP device
            model(img=[img], img_metas=[metas], return_loss=False)
        with pytest.raises(RuntimeError):
            model.backbone(img)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'mcgaze_hip.h')).read()
    declared = set(re.findall(r'\b(mcg_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'mcg_stream', 'mcg_dtype'}
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.mcg_abi_version() == L.ABI_VERSION == int(re.search(r"#define MCG_ABI_VERSION (\d+)", hdr).group(1))
    # the binding's weight-table keys are the header's enums, in order
    sw = re.findall(r'MCG_SW_([A-Z0-9_]+)', hdr[hdr.index('MCG_SW_IN_PROJ_W = 0'):hdr.index('MCG_SW_COUNT')])
    gw = re.findall(r'MCG_GW_([A-Z0-9_]+)', hdr[hdr.index('MCG_GW_FC_W = 0'):hdr.index('MCG_GW_COUNT')])
    assert sw == L.STAGE_KEYS and gw == L.GAZE_KEYS and L.SW_COUNT == 39 and L.GW_COUNT == 7


@pytest.mark.parametrize('epc', [4, 8])
def test_dyn_permutation_matches_reference_view(epc):
    """params[:, :d*f].view(d, f) / params[:, -d*f:].view(f, d) (transformer.py:1134-1137) -> the MFMA-fragment-major order dynconv_kernel
    reads (packing.dyn_permutation): every element of param_in^T [f][d] and param_out^T [d][f] sits where lane l of the wave-wide 16-byte
    load of (tile, chunk pair j) expects it, and the permutation is a bijection."""
    from mcgaze_amd.packing import dyn_permutation
    d, f = 256, 64
    theta = torch.arange(2 * d * f, dtype=torch.float32)
    w_in, w_out = theta[:d * f].view(d, f), theta[-d * f:].view(f, d)
    perm = dyn_permutation(d, f, epc)
    assert sorted(perm.tolist()) == list(range(2 * d * f))
    p = theta[perm]
    win_t, wout_t = w_in.t(), w_out.t()                                # [f][d], [d][f]: B operands with K contiguous
    pin = p[:d * f].view(f // 32, d // (2 * epc), 64, epc)
    pout = p[d * f:].view(d // 32, f // (2 * epc), 64, epc)
    for lane in (0, 5, 31, 32, 63):
        for t, j in ((0, 0), (1, 3), (f // 32 - 1, d // (2 * epc) - 1)):
            k0 = (2 * j + (lane >> 5)) * epc
            assert torch.equal(pin[t, j, lane], win_t[32 * t + (lane & 31), k0:k0 + epc])
        for t, j in ((0, 0), (5, 1), (d // 32 - 1, f // (2 * epc) - 1)):
            k0 = (2 * j + (lane >> 5)) * epc
            assert torch.equal(pout[t, j, lane], wout_t[32 * t + (lane & 31), k0:k0 + epc])


def test_bn_fold_equals_conv_then_bn():
    from mcgaze_amd.packing import fold_bn
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(1).items() if k.startswith('backbone.layer1.0.')}
    x = torch.randn(2, 64, 9, 9)
    w, b = fold_bn(sd, 'backbone.layer1.0.conv1.weight', 'backbone.layer1.0.bn1')
    ref = torch.nn.functional.batch_norm(torch.nn.functional.conv2d(x, sd['backbone.layer1.0.conv1.weight']),
                                         sd['backbone.layer1.0.bn1.running_mean'], sd['backbone.layer1.0.bn1.running_var'],
                                         sd['backbone.layer1.0.bn1.weight'], sd['backbone.layer1.0.bn1.bias'], False, 0., 1e-5)
    assert torch.allclose(torch.nn.functional.conv2d(x, w, b), ref, atol=1e-5)


def test_checkpoint_ingestion_envelope_and_key_rewrite(tmp_path):
    """the mmcv ``.pth`` envelope (``meta`` + ``state_dict``), the DataParallel ``module.`` prefix
    stripped (mmdet/apis/inference.py:45), ``meta['CLASSES']`` taken over (:46-53), extra / missing keys tolerated with a
    warning (mmcv load_checkpoint is non-strict), a bare state_dict accepted, a non-dict rejected."""
    from mcgaze_amd import init_detector
    from mcgaze_amd.apis import load_checkpoint
    ref = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(5).items()}
    env = dict(meta=dict(CLASSES=('a', 'b', 'c'), mmdet_version='2.x'), state_dict={'module.' + k: v for k, v in ref.items()},
               optimizer=dict(state={}))
    env['state_dict']['module.roi_head.some_dead_head.weight'] = torch.zeros(3)      # unexpected key
    dropped = 'roi_head.bbox_head.0.fc_cls.weight'                                    # dead at inference
    del env['state_dict']['module.' + dropped]
    path = str(tmp_path / 'ckpt.pth')
    torch.save(env, path)
    with pytest.warns(UserWarning, match='missing keys'):
        model = init_detector(OWN_CFG, path, device='cpu')
    assert model.CLASSES == ('a', 'b', 'c') and model.cfg.clip_length == 7 and not model.training
    sd = model.state_dict()
    for k, v in ref.items():
        if k != dropped:
            assert torch.equal(sd[k], v), k
    bare = str(tmp_path / 'bare.pth')
    torch.save(ref, bare)
    _, m2 = build(OWN_CFG)
    load_checkpoint(m2, bare, strict=True)
    assert torch.equal(m2.state_dict()[dropped], ref[dropped])
    torch.save([1, 2, 3], bare)
    with pytest.raises(RuntimeError, match='No state_dict'):
        load_checkpoint(m2, bare)


def test_split_pack_layout_and_accuracy():
    """packing.split_pack (the MCG_F16X3 weight operand): per 8 K elements 8 fp16 high parts then 8 fp16 low parts; hi + lo
    reproduces the f32 value to 2^-21 relative where the low part is a normal half, to 6e-8 absolute below that, and saturates
    per half beyond +-65504."""
    from mcgaze_amd.packing import split_pack
    g = torch.Generator().manual_seed(5)
    w = torch.randn(6, 64, generator=g) * torch.logspace(-6, 3, 64)[None, :]
    p = split_pack(w)
    assert p.dtype == torch.float16 and tuple(p.shape) == (6, 128)
    v = p.reshape(6, 8, 2, 8).float()
    hi, lo = v[:, :, 0, :].reshape(6, 64), v[:, :, 1, :].reshape(6, 64)
    assert torch.equal(hi, w.to(torch.float16).float())
    assert torch.equal(lo, (w - hi).to(torch.float16).float())
    err = (hi + lo - w).abs()
    assert (err <= torch.maximum(w.abs() * 2.0 ** -21, torch.tensor(2.0 ** -24))).all()
    big = split_pack(torch.tensor([[1e6, -7e4, 65504.0, 3e-9, 0.0, 1.0, -1.0, 100000.0]])).float().reshape(2, 8)
    assert torch.isfinite(big).all() and big[0, 0] == 65504.0 and big[1, 0] == 65504.0 and big[0, 3] == 0.0
