"""test-time preprocessing.  CPU part: the geometry / RNG bookkeeping of mcgaze_amd/pipeline.py and
the oracle against goldens captured from the reference's own transform classes (tests/golden/preprocess_kat.json,
oracle/dev/make_preprocess_goldens.py) and hand-computed pixel cases of the restated cv2 arithmetic (parity unpinned at the
mmcv / OpenCV boundary -- see oracle/preprocess_oracle.py).  GPU part: mcg_preprocess_frames bit-exact against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from mcgaze_amd import Config
from mcgaze_amd import lib as L
from mcgaze_amd import pipeline as P
from oracle import preprocess_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'preprocess_kat.json')))
OWN_CFG = os.path.join(ROOT, 'configs', 'mcgaze', 'r50_clip7_gaze360.py')
NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
GAZE360 = [dict(type='LoadImageFromFile'), dict(type='CenterCrop', crop_size=(0.68, 0.68), crop_type='relative_range'),
           dict(type='Resize', img_scale=(224, 224), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.0),
           dict(type='Normalize', **NORM), dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'),
           dict(type='Collect', keys=['img'])]
L2CS = [dict(type='LoadImageFromFile'), dict(type='Resize', img_scale=(448, 448), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.0),
        dict(type='Normalize', **NORM), dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img'])]


def pattern(shape):
    """The synthetic frame of oracle/dev/make_preprocess_goldens.py."""
    return ((np.arange(shape[0])[:, None, None] * 7 + np.arange(shape[1])[None, :, None] * 3 + np.arange(3)[None, None, :] * 50) % 256).astype(np.uint8)


def test_own_config_names_the_reference_test_pipeline():
    cfg = Config.fromfile(OWN_CFG)
    kinds = [t['type'] for t in cfg.data.test.pipeline]
    assert kinds == [t['type'] for t in GAZE360]
    P.DevicePipeline(cfg.data.test.pipeline)
    # L2CS setting: same model through `_base_`, test pipeline replaced (`_delete_`), no crop, long side 448
    l2 = Config.fromfile(os.path.join(ROOT, 'configs', 'mcgaze', 'r50_clip7_l2cs.py'))
    assert l2.model == cfg.model and '_delete_' not in l2.data.test
    assert [dict(t) for t in l2.data.test.pipeline] == L2CS and l2.data.test.img_prefix == 'data/l2cs/test_rawframes/'
    assert l2.data.workers_per_gpu == cfg.data.workers_per_gpu   # inherited key survives the merge
    P.DevicePipeline(l2.data.test.pipeline)


@pytest.mark.parametrize('case', KAT['random_cases'], ids=lambda c: f"seed{c['seed']}")
def test_crop_draws_windows_and_meta_match_reference_classes(case):
    rng = np.random.RandomState(case['seed'])   # the reference draws from the global RNG seeded the same way
    pipe = P.DevicePipeline(GAZE360)
    for g in case['frames']:
        shape = tuple(g['ori_shape'])
        p = pipe.plan(shape, rng)
        y, x, h, w = p.crop
        assert [h, w, 3] == g['crop_shape']
        assert int(pattern(shape)[y:y + h, x:x + w].astype(np.int64).sum()) == g['crop_window_sum']   # same window, not just same size
        assert list(p.img_shape) == g['img_shape'] and list(p.pad_shape) == g['pad_shape']
        assert p.scale_factor.dtype == np.float32 and [float(v) for v in p.scale_factor] == g['scale_factor']
        assert p.flip is g['flip'] and p.flip_direction is g['flip_direction']
        assert [float(v) for v in p.img_norm_cfg['mean']] == g['mean'] and [float(v) for v in p.img_norm_cfg['std']] == g['std']
        meta = pipe.collect.meta(p)
        assert list(meta) == ['filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip', 'flip_direction', 'img_norm_cfg']
    assert float(rng.random_sample()) == case['next_uniform']   # consumed exactly the reference's draws (crop + flip per frame)


def test_oracle_geometry_matches_reference_classes():
    for case in KAT['random_cases']:
        rng = np.random.RandomState(case['seed'])
        for g in case['frames']:
            u = rng.rand(1)[0]
            rng.random_sample()
            h0, w0 = g['ori_shape'][:2]
            ch, cw = po.center_crop_size(h0, w0, (0.68, 0.68), 'relative_range', u)
            y, x, ch, cw = po.center_crop_window(h0, w0, ch, cw)
            assert [ch, cw, 3] == g['crop_shape']
            nw, nh = po.rescale_size(cw, ch, (224, 224))
            assert [nh, nw, 3] == g['img_shape']
    for g in KAT['deterministic_crops']:
        h0, w0 = g['ori_shape'][:2]
        ch, cw = po.center_crop_size(h0, w0, tuple(g['crop_size']), g['crop_type'])
        assert list(po.center_crop_window(h0, w0, ch, cw)[2:]) == g['crop_shape'][:2]
        p = P.DevicePipeline([dict(type='LoadImageFromFile'), dict(type='CenterCrop', crop_size=tuple(g['crop_size']), crop_type=g['crop_type']),
                              dict(type='Normalize', **NORM), dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img'])]).plan(tuple(g['ori_shape']))
        assert list(p.img_shape) == g['crop_shape']
    pipe = P.DevicePipeline(L2CS)
    for g in KAT['l2cs']:
        p = pipe.plan(tuple(g['ori_shape']), np.random.RandomState(0))
        assert list(p.img_shape) == g['img_shape'] and list(p.pad_shape) == g['pad_shape'] and [float(v) for v in p.scale_factor] == g['scale_factor']


def test_resize_hand_cases():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (6, 8, 3)).astype(np.uint8)
    assert np.array_equal(po.resize_linear_u8(img, 8, 6), img)                                   # identity
    box = (img.astype(np.int64).reshape(3, 2, 4, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(po.resize_linear_u8(img, 4, 3), box)                                   # exact 2x down = 2x2 box mean, rounded
    assert np.array_equal(po.resize_linear_u8(np.full((5, 7, 3), 200, np.uint8), 13, 9), np.full((9, 13, 3), 200, np.uint8))
    ramp = np.array([[[0] * 3, [255] * 3]], dtype=np.uint8)                                      # 1 x 2 -> 1 x 4
    # centres at -0.25, 0.25, 0.75, 1.25 -> weights of the right pixel 0, 512/2048, 1536/2048, clamp: 0, (255*512>>4)*2048>>16 = 255 -> (255+2)>>2 = 64, 191, 255
    assert po.resize_linear_u8(ramp, 4, 1)[0, :, 0].tolist() == [0, 64, 191, 255]
    assert po.resize_linear_u8(ramp.transpose(1, 0, 2), 1, 4)[:, 0, 0].tolist() == [0, 64, 191, 255]
    one = np.array([[[9, 8, 7]]], dtype=np.uint8)
    assert np.array_equal(po.resize_linear_u8(one, 3, 2), np.broadcast_to(one, (2, 3, 3)))      # single source pixel


def test_resize_fixed_point_cases_worked_by_hand():
    """cv2.resize(INTER_LINEAR) on 8-bit pixels, numbers worked out by hand from OpenCV's published fixed-point path (resize.cpp:
    source coordinate (d + .5) * src/dst - .5, coefficients rint(f * 2048), horizontal pass in ints, vertical pass
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2) -- up-scale, down-scale by a non-integer factor, odd sizes.
    Every scale is a dyadic rational so that the coordinates are exact and the derivation does not depend on double rounding."""
    # 1 x 2 -> 1 x 8 (scale 1/4): coordinates -.375 -.125 .125 .375 .625 .875 1.125 1.375 -> right-pixel coefficients 0 0 256 768 1280 1792
    # then clamped (s >= src - 1 -> f = 0).  For [0, 255]: 255 * 256 = 65280 -> >> 4 = 4080 -> * 2048 >> 16 = 127 -> (127 + 2) >> 2 = 32;
    # 195840 -> 12240 -> 382 -> 96;  326400 -> 20400 -> 637 -> 159;  456960 -> 28560 -> 892 -> 223
    ramp = np.array([[[0] * 3, [255] * 3]], dtype=np.uint8)
    assert po.resize_linear_u8(ramp, 8, 1)[0, :, 1].tolist() == [0, 0, 32, 96, 159, 223, 255, 255]
    # 1 x 6 -> 1 x 4 (scale 3/2): coordinates .25 1.75 3.25 4.75 -> (s, f) = (0, .25) (1, .75) (3, .25) (4, .75); coefficients 1536 / 512
    #   [1, 2, 4, 7, 11, 16]: 1*1536 + 2*512 = 2560 -> 160 -> 5 -> (5+2)>>2 = 1        (exact 1.25)
    #                         2*512 + 4*1536 = 7168 -> 448 -> 14 -> 16>>2 = 4          (exact 3.5: rounds up)
    #                         7*1536 + 11*512 = 16384 -> 1024 -> 32 -> 34>>2 = 8       (exact 8.0)
    #                         11*512 + 16*1536 = 30208 -> 1888 -> 59 -> 61>>2 = 15     (exact 14.75)
    row = np.array([[[v] * 3 for v in (1, 2, 4, 7, 11, 16)]], dtype=np.uint8)
    assert po.resize_linear_u8(row, 4, 1)[0, :, 2].tolist() == [1, 4, 8, 15]
    assert po.resize_linear_u8(row.transpose(1, 0, 2), 1, 4)[:, 0, 0].tolist() == [1, 4, 8, 15]          # the same along y
    row = np.array([[[v] * 3 for v in (0, 40, 80, 120, 160, 200)]], dtype=np.uint8)
    assert po.resize_linear_u8(row, 4, 1)[0, :, 0].tolist() == [10, 70, 130, 190]
    # 3 x 3 -> 6 x 6 (odd source, scale 1/2), v[y][x] = 10 y + 3 x: output (1, 1) has (s, f) = (0, .25) on both axes:
    #   row 0: 0*1536 + 3*512 = 1536;  row 1: 10*1536 + 13*512 = 22016;  vertical: (1536 * (1536 >> 4)) >> 16 = 2,
    #   (512 * (22016 >> 4)) >> 16 = 10  ->  (2 + 10 + 2) >> 2 = 3            (exact 3.25)
    # output (4, 2): y (s, f) = (1, .75), x (s, f) = (0, .75): row 1: 10*512 + 13*1536 = 25088, row 2: 20*512 + 23*1536 = 45568;
    #   (512 * (25088 >> 4)) >> 16 = 12 (12.25), (1536 * (45568 >> 4)) >> 16 = 66 (66.75) -> (12 + 66 + 2) >> 2 = 20   (exact 19.75)
    # corners replicate: output (0, 0) = v[0][0] = 0, output (5, 5) = v[2][2] = 26
    v = np.array([[[10 * y + 3 * x] * 3 for x in range(3)] for y in range(3)], dtype=np.uint8)
    out = po.resize_linear_u8(v, 6, 6)
    assert (int(out[1, 1, 0]), int(out[4, 2, 0]), int(out[0, 0, 0]), int(out[5, 5, 0])) == (3, 20, 0, 26)
    # 5 x 7 -> 10 x 14 and back keeps a constant image constant (no drift from the +2 >> 2 rounding)
    c = np.full((5, 7, 3), 77, np.uint8)
    assert np.array_equal(po.resize_linear_u8(po.resize_linear_u8(c, 14, 10), 7, 5), c)


def test_normalize_pad_format_hand_cases():
    img = np.zeros((2, 3, 3), np.uint8)
    img[0, 0] = (10, 20, 30)   # B, G, R
    out = po.imnormalize(img, **NORM)
    m, s = np.float32(NORM['mean']), np.float32(NORM['std'])
    want = [(np.float32(30) - m[0]) * np.float32(1.0 / np.float64(s[0])), (np.float32(20) - m[1]) * np.float32(1.0 / np.float64(s[1])),
            (np.float32(10) - m[2]) * np.float32(1.0 / np.float64(s[2]))]
    assert out.dtype == np.float32 and out[0, 0].tolist() == [float(v) for v in want]
    padded = po.impad_to_multiple(out, 32)
    assert padded.shape == (32, 32, 3) and np.array_equal(padded[:2, :3], out) and padded[2:].sum() == 0 and padded[:, 3:].sum() == 0
    chw, meta = po.test_pipeline(pattern((95, 143, 3)), u=0.5)
    assert chw.shape == (3, meta['pad_shape'][0], meta['pad_shape'][1]) and chw.flags['C_CONTIGUOUS']
    assert np.array_equal(po.collate_clip([chw, chw[:, :32, :64]])[1, :, :32, :64], chw[:, :32, :64])


def test_pipeline_argument_errors_and_no_cpu_pixel_path():
    with pytest.raises(ValueError, match='Invalid crop_type'):
        P.CenterCrop((0.5, 0.5), crop_type='nope')
    with pytest.raises(AssertionError):
        P.CenterCrop((0.5, 1.5), crop_type='relative')
    with pytest.raises(NotImplementedError):
        P.RandomFlip(flip_ratio=0.5)
    with pytest.raises(ValueError, match='flip_ratios must be'):
        P.RandomFlip(flip_ratio='x')
    with pytest.raises(NotImplementedError):
        P.Pad(size_divisor=32, pad_val=dict(img=114))
    with pytest.raises(AssertionError):
        P.Pad()
    with pytest.raises(KeyError):
        P.DevicePipeline([dict(type='LoadImageFromFile'), dict(type='Mosaic')])
    with pytest.raises(ValueError, match='must start with LoadImageFromFile'):
        P.DevicePipeline(GAZE360[1:])
    with pytest.raises(L.McgError, match='no CPU path'):
        P.DevicePipeline(GAZE360)([pattern((64, 64, 3))], device='cpu')


# ------------------------------------------------------------------------------------------------ GPU
def oracle_clip(frames, us, chain):
    outs, metas = [], []
    for f, u in zip(frames, us):
        if chain is GAZE360:
            o, m = po.test_pipeline(f, u=u)
        else:
            o, m = po.test_pipeline(f, crop=None, img_scale=(448, 448))
        outs.append(o)
        metas.append(m)
    return po.collate_clip(outs), metas


@pytest.mark.gpu
@pytest.mark.parametrize('chain', [GAZE360, L2CS], ids=['gaze360', 'l2cs'])
def test_device_preprocess_is_bit_exact_against_oracle(chain):
    rs = np.random.RandomState(11)
    shapes = [(720, 1280, 3), (224, 224, 3), (301, 257, 3), (95, 143, 3), (333, 37, 3), (448, 300, 3), (64, 64, 3)]
    frames = [rs.randint(0, 256, s).astype(np.uint8) for s in shapes]
    seed = 5
    rng = np.random.RandomState(seed)
    us = []
    for _ in frames:
        us.append(rng.rand(1)[0] if chain is GAZE360 else None)
        rng.random_sample()
    want, want_metas = oracle_clip(frames, us, chain)
    img, metas = P.DevicePipeline(chain)(frames, device='cuda:0', rng=np.random.RandomState(seed))
    torch.cuda.synchronize()
    got = img.cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.array_equal(got, want)      # integer resize + float32 normalise: bit for bit
    for m, w in zip(metas, want_metas):
        assert m['img_shape'] == w['img_shape'] and m['pad_shape'] == w['pad_shape'] and m['ori_shape'] == w['ori_shape']
        assert np.array_equal(m['scale_factor'], w['scale_factor']) and m['flip'] is False


@pytest.mark.gpu
def test_device_preprocess_from_files_and_into_the_model(tmp_path):
    from PIL import Image
    from mcgaze_amd import init_detector
    rs = np.random.RandomState(3)
    frames = [rs.randint(0, 256, (260, 260, 3)).astype(np.uint8) for _ in range(7)]
    names = []
    for i, f in enumerate(frames):
        names.append(f'{i:06d}.png')
        Image.fromarray(f[..., ::-1]).save(str(tmp_path / names[-1]))     # PIL writes RGB; the pipeline decodes back to BGR
    pipe = P.DevicePipeline(Config.fromfile(OWN_CFG).data.test.pipeline)
    a, metas = pipe(names, device='cuda:0', rng=np.random.RandomState(9), img_prefix=str(tmp_path))
    b, _ = pipe(frames, device='cuda:0', rng=np.random.RandomState(9))
    assert torch.equal(a, b) and metas[3]['ori_filename'] == names[3] and metas[3]['filename'] == os.path.join(str(tmp_path), names[3])
    assert tuple(a.shape) == (7, 3, 224, 224)
    model = init_detector(OWN_CFG, None, device='cuda:0', precision='fp32')
    (boxes, labels), gaze = model(img=[a], img_metas=[metas], return_loss=False, rescale=True, format=False)
    assert len(boxes) == 7 and tuple(gaze['gaze_score'].shape) == (7, 3) and torch.isfinite(gaze['gaze_score']).all()
    assert pipe(names[:0], device='cuda:0')[0].shape[0] == 0
