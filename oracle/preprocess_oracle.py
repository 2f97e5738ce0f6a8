"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's test-time preprocessing.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path
(mcgaze_amd/pipeline.py -> mcg_preprocess_frames in libmcgaze_hip.so) never does.

Restates the eight transforms of configs/_base_/datasets/gaze360.py:27-36 (and the L2CS variant without the crop,
configs/multiclue_gaze/multiclue_gaze_r50_l2cs.py:31-39) for ONE frame:

  LoadImageFromFile     BGR uint8 HWC (mmdet/datasets/pipelines/loading.py:52-74)            -> caller supplies the array
  CenterCrop            mmdet/datasets/pipelines/transforms.py:1036-1052 (centred window), :1101-1130 (size draw)
  Resize(keep_ratio)    transforms.py:216-242 -> mmcv.imrescale -> cv2.resize(INTER_LINEAR)
  RandomFlip(0.0)       transforms.py:463-530: flip_ratio 0 never flips (but draws one uniform); flip=False, flip_direction=None
  Normalize(to_rgb)     transforms.py:739-755 -> mmcv.imnormalize
  Pad(size_divisor)     transforms.py:665-683 -> mmcv.impad_to_multiple (zeros, bottom/right)
  DefaultFormatBundle   mmdet/datasets/pipelines/formatting.py:229-231: HWC -> CHW, contiguous
  Collect(['img'])      formatting.py:279-347: meta keys filename, ori_filename, ori_shape, img_shape, pad_shape,
                        scale_factor, flip, flip_direction, img_norm_cfg

PARITY PINNING.  CenterCrop is pure numpy inside the reference tree and IS pinned: tests/golden/preprocess_kat.json was
captured by running the reference's own class (oracle/dev/make_preprocess_goldens.py).  The pixel arithmetic of Resize,
Normalize and Pad lives in third-party code that is neither under /root/reference nor installable here -- mmcv-full 1.4.8
(`mmcv.imrescale`, `mmcv.imnormalize`, `mmcv.impad_to_multiple`) on top of OpenCV (`cv2.resize`, `cv2.cvtColor`,
`cv2.subtract`, `cv2.multiply`) -- and the reference holds no test or fixture for it: **parity unpinned** at that boundary.
What is restated below is their published algorithm:
  * mmcv.image.geometric.rescale_size / _scale_size: scale = min(long_edge / max(h, w), short_edge / min(h, w)),
    new (w, h) = int(w * scale + 0.5), int(h * scale + 0.5);
  * OpenCV imgproc resize.cpp, INTER_LINEAR on 8-bit: source coordinate (d + 0.5) * (src / dst) - 0.5 computed in double then
    cast to float, floor + fraction, clamp to the image with zero fraction at the borders, coefficients rounded to 11-bit
    fixed point (saturate_cast<short>(f * 2048), round-half-even), horizontal pass in 32-bit ints, vertical pass
    uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
  * mmcv.imnormalize: float32 copy, BGR->RGB, subtract mean, multiply by 1/std -- in float32 (OpenCV converts a scalar
    operand to the working type of a 32F array).
"""
import os

import numpy as np

COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS


def center_crop_size(h, w, crop_size, crop_type, u=None):
    """transforms.py:1101-1130.  `u` replaces the reference's np.random.rand(1) draw for 'relative_range'."""
    if crop_type == 'absolute':
        return min(crop_size[0], h), min(crop_size[1], w)
    if crop_type == 'relative':
        return int(h * crop_size[0] + 0.5), int(w * crop_size[1] + 0.5)
    if crop_type == 'relative_range':
        cs = np.asarray(crop_size, dtype=np.float32)
        crop_h, crop_w = cs + np.asarray([u], dtype=np.float64) * (1 - cs)   # float32 array op float64 array -> float64
        return int(h * crop_h + 0.5), int(w * crop_w + 0.5)
    raise ValueError(f'Invalid crop_type {crop_type}.')


def center_crop_window(h, w, crop_h, crop_w):
    """transforms.py:1036-1045: the window is centred (the random offsets are commented out upstream)."""
    margin_h, margin_w = max(h - crop_h, 0), max(w - crop_w, 0)
    y0, x0 = int(margin_h / 2 + 0.5), int(margin_w / 2 + 0.5)
    return y0, x0, min(crop_h, h - y0), min(crop_w, w - x0)   # numpy slicing clips at the border


def rescale_size(w, h, scale):
    """mmcv.image.geometric.rescale_size for a tuple scale (img_scale in the configs)."""
    long_edge, short_edge = max(scale), min(scale)
    f = min(long_edge / max(h, w), short_edge / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def _linear_coeffs(dst, src):
    """OpenCV resize.cpp: per destination index the left source index and the two 11-bit coefficients."""
    scale = 1.0 / (float(dst) / float(src))          # scale_x = 1. / inv_scale_x, double
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0
    s[hi] = src - 1
    a1 = np.rint(f * np.float32(COEF_ONE)).astype(np.int64)                    # saturate_cast<short>: round half to even
    a0 = np.rint((np.float32(1) - f) * np.float32(COEF_ONE)).astype(np.int64)
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, a0, a1


def resize_linear_u8(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 HWC, restated (see header)."""
    h, w = img.shape[:2]
    x0, x1, ax0, ax1 = _linear_coeffs(new_w, w)
    y0, y1, by0, by1 = _linear_coeffs(new_h, h)
    src = img.astype(np.int64)
    hor = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]    # [h, new_w, c], scaled by 2^11
    r0, r1 = hor[y0], hor[y1]
    out = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def imnormalize(img_u8_bgr, mean, std, to_rgb=True):
    img = img_u8_bgr.astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    # Normalize keeps mean / std as float32 arrays (transforms.py:735-736); mmcv.imnormalize widens them to float64 and takes
    # 1 / std there; OpenCV then narrows a scalar operand of a 32F array back to float
    m = np.asarray(mean, dtype=np.float32)
    sinv = (1.0 / np.asarray(std, dtype=np.float32).astype(np.float64)).astype(np.float32)
    return ((img - m).astype(np.float32) * sinv).astype(np.float32)


def impad_to_multiple(img, divisor, pad_val=0):
    h, w = img.shape[:2]
    ph, pw = int(np.ceil(h / divisor)) * divisor, int(np.ceil(w / divisor)) * divisor
    out = np.full((ph, pw) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:h, :w] = img
    return out


def test_pipeline(img_bgr, u=None, crop=(0.68, 0.68), crop_type='relative_range', img_scale=(224, 224),
                  mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True, size_divisor=32, filename=None):
    """One frame through the eight transforms.  crop=None skips CenterCrop (the L2CS config).  Returns (CHW float32, meta)."""
    ori_shape = tuple(img_bgr.shape)
    img = img_bgr
    if crop is not None:
        ch, cw = center_crop_size(img.shape[0], img.shape[1], crop, crop_type, u)
        y0, x0, ch, cw = center_crop_window(img.shape[0], img.shape[1], ch, cw)
        img = img[y0:y0 + ch, x0:x0 + cw]
    h, w = img.shape[:2]
    new_w, new_h = rescale_size(w, h, img_scale)
    img = resize_linear_u8(img, new_w, new_h)
    scale_factor = np.array([new_w / w, new_h / h, new_w / w, new_h / h], dtype=np.float32)
    img_shape = tuple(img.shape)
    img = imnormalize(img, mean, std, to_rgb)
    img = impad_to_multiple(img, size_divisor, 0)
    meta = dict(filename=filename, ori_filename=filename if filename is None else os.path.basename(filename), ori_shape=ori_shape, img_shape=img_shape, pad_shape=tuple(img.shape),
                scale_factor=scale_factor, flip=False, flip_direction=None,
                img_norm_cfg=dict(mean=np.array(mean, dtype=np.float32), std=np.array(std, dtype=np.float32), to_rgb=to_rgb))
    return np.ascontiguousarray(img.transpose(2, 0, 1)), meta


def collate_clip(frames):
    """mmcv.parallel.collate for stack=True DataContainers: zero-pad bottom/right to the largest H, W of the batch."""
    H, W = max(f.shape[1] for f in frames), max(f.shape[2] for f in frames)
    out = np.zeros((len(frames), frames[0].shape[0], H, W), dtype=np.float32)
    for i, f in enumerate(frames):
        out[i, :, :f.shape[1], :f.shape[2]] = f
    return out
