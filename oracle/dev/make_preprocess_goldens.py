"""DEV-ONLY: goldens for SURVEY.md section 8(f)-3 (test-time preprocessing geometry).

Runs the reference's OWN transform classes in the build container (imported through the mmcv stand-in):
  * CenterCrop + RandomFlip(0.0) of mmdet/datasets/pipelines/transforms.py on dummy frames with a seeded GLOBAL numpy
    RNG, in the order of configs/_base_/datasets/gaze360.py:27-36 -- pins the crop-size draw, the centred window and the
    RNG consumption per frame (both classes are pure numpy when nothing is flipped);
  * Resize(keep_ratio) + Pad + Normalize bookkeeping (img_shape / pad_shape / scale_factor / img_norm_cfg) with the three
    third-party pixel functions they delegate to -- mmcv.imrescale, mmcv.imnormalize, mmcv.impad_to_multiple, absent here --
    replaced by oracle/preprocess_oracle.py's restatement.  That pins the bookkeeping, NOT the pixel arithmetic, which stays
    "parity unpinned" (see the oracle's header).
Writes tests/golden/preprocess_kat.json.
"""
import json
import os
import sys
import warnings

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import mmcv_standin  # noqa: E402,F401

sys.path.insert(0, '/root/reference')
import numpy as np  # noqa: E402
import mmcv  # noqa: E402  (the stand-in)

from oracle import preprocess_oracle as po  # noqa: E402


def _imrescale(img, scale, return_scale=False, interpolation='bilinear', backend=None):
    h, w = img.shape[:2]
    new_w, new_h = po.rescale_size(w, h, scale)
    out = po.resize_linear_u8(img, new_w, new_h)
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return (out, f) if return_scale else out


mmcv.imrescale = _imrescale
mmcv.imnormalize = lambda img, mean, std, to_rgb=True: po.imnormalize(img, mean, std, to_rgb)
mmcv.impad_to_multiple = lambda img, divisor, pad_val=0: po.impad_to_multiple(img, divisor, pad_val)

from mmdet.datasets.pipelines.transforms import CenterCrop, Normalize, Pad, RandomFlip, Resize  # noqa: E402

SHAPES = [(720, 1280, 3), (224, 224, 3), (301, 257, 3), (95, 143, 3), (1000, 37, 3), (448, 300, 3)]
cases = []
for seed in (0, 1, 7):
    np.random.seed(seed)
    crop = CenterCrop(crop_size=(0.68, 0.68), crop_type='relative_range')
    flip = RandomFlip(flip_ratio=0.0)
    resize = Resize(img_scale=(224, 224), keep_ratio=True)
    norm = Normalize(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    pad = Pad(size_divisor=32)
    frames = []
    for i, shp in enumerate(SHAPES * 2):
        img = np.zeros(shp, dtype=np.uint8)
        img[...] = (np.arange(shp[0])[:, None, None] * 7 + np.arange(shp[1])[None, :, None] * 3 + np.arange(3)[None, None, :] * 50) % 256
        r = dict(img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=['img'])
        r = crop(r)
        window_sum = int(r['img'].astype(np.int64).sum())
        crop_shape = tuple(int(v) for v in r['img_shape'])
        r = resize(r)
        r = flip(r)
        r = norm(r)
        r = pad(r)
        frames.append(dict(ori_shape=list(shp), crop_shape=list(crop_shape), crop_window_sum=window_sum,
                           img_shape=[int(v) for v in r['img_shape']], pad_shape=[int(v) for v in r['pad_shape']],
                           scale_factor=[float(v) for v in r['scale_factor']], flip=bool(r['flip']),
                           flip_direction=r['flip_direction'], to_rgb=bool(r['img_norm_cfg']['to_rgb']),
                           mean=[float(v) for v in r['img_norm_cfg']['mean']], std=[float(v) for v in r['img_norm_cfg']['std']]))
    cases.append(dict(seed=seed, frames=frames, next_uniform=float(np.random.random_sample())))

# other crop types (deterministic)
det = []
for ctype, csize in (('relative', (0.5, 0.8)), ('absolute', (100, 150))):
    c = CenterCrop(crop_size=csize, crop_type=ctype)
    for shp in SHAPES:
        img = np.zeros(shp, dtype=np.uint8)
        r = c(dict(img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=['img']))
        det.append(dict(crop_type=ctype, crop_size=list(csize), ori_shape=list(shp), crop_shape=[int(v) for v in r['img_shape']]))
# the L2CS chain: no crop, Resize((448, 448), keep_ratio) -> Pad(32)
l2cs = []
resize = Resize(img_scale=(448, 448), keep_ratio=True)
pad = Pad(size_divisor=32)
for shp in SHAPES:
    img = np.zeros(shp, dtype=np.uint8)
    r = pad(resize(dict(img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=['img'])))
    l2cs.append(dict(ori_shape=list(shp), img_shape=[int(v) for v in r['img_shape']], pad_shape=[int(v) for v in r['pad_shape']],
                     scale_factor=[float(v) for v in r['scale_factor']]))
out = os.path.join(ROOT, 'tests', 'golden', 'preprocess_kat.json')
json.dump(dict(generator='oracle/dev/make_preprocess_goldens.py (reference transforms.py classes, seeded global numpy RNG)',
               random_cases=cases, deterministic_crops=det, l2cs=l2cs), open(out, 'w'), indent=1)
print('wrote', out, len(cases), 'seeded cases')
