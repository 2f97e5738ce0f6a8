"""Seeded synthetic weights and clips for the MCGaze per-clip forward path.

There is no checkpoint or dataset in the build container, so tests,
``bench.py`` and the golden-vector generator all use the generators in this file:
random-init weights with the architecture's exact ``state_dict`` layout (744 tensors,
SURVEY.md §8(a) "State-dict layout") and N(0,1) clips in the normalised-pixel domain.

Every tensor is drawn from its own ``numpy.random.RandomState`` keyed by the tensor
name, so the values do not depend on iteration order or on which subset is requested.
BN statistics and affine terms are randomised (not the identity) so that a wrong
BN fold, a transposed weight or a swapped clue token changes the output.
"""
import zlib

import numpy as np

ARCH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _rs(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def _normal(name, seed, shape, std):
    return (_rs(name, seed).standard_normal(shape) * std).astype(np.float32)


def _uniform(name, seed, shape, lo, hi):
    return _rs(name, seed).uniform(lo, hi, shape).astype(np.float32)


def _bn(sd, prefix, c, seed, gamma=(0.5, 1.5)):
    sd[prefix + '.weight'] = _uniform(prefix + '.weight', seed, (c,), *gamma)
    sd[prefix + '.bias'] = _normal(prefix + '.bias', seed, (c,), 0.1)
    sd[prefix + '.running_mean'] = _normal(prefix + '.running_mean', seed, (c,), 0.1)
    sd[prefix + '.running_var'] = _uniform(prefix + '.running_var', seed, (c,), 0.5, 1.5)
    sd[prefix + '.num_batches_tracked'] = np.array(0, dtype=np.int64)


def _loguniform(name, seed, shape, lo, hi):
    return np.exp(_rs(name, seed).uniform(np.log(lo), np.log(hi), shape)).astype(np.float32)


def _bn_trained(sd, prefix, conv_key, seed, gamma=(0.03, 3.0)):
    """BN statistics of the 'trained' weight family: running_var log-uniform over three decades with the conv's ROWS scaled by its square
    root (so the layer stays calibrated, as a trained BN keeps it: unit variance after normalisation), gamma log-uniform over two decades
    (a trained net's mix of strong and nearly dead channels) -- the folded weights w * gamma / sigma then span two decades per matrix."""
    c = sd[conv_key].shape[0]
    var = _loguniform(prefix + '.running_var', seed, (c,), 0.01, 10.0)
    sd[conv_key] = (sd[conv_key] * np.sqrt(var)[:, None, None, None]).astype(np.float32)
    sd[prefix + '.weight'] = _loguniform(prefix + '.weight', seed, (c,), *gamma)
    sd[prefix + '.bias'] = _normal(prefix + '.bias', seed, (c,), 0.1)
    sd[prefix + '.running_mean'] = (_normal(prefix + '.running_mean', seed, (c,), 0.1) * np.sqrt(var)).astype(np.float32)
    sd[prefix + '.running_var'] = var
    sd[prefix + '.num_batches_tracked'] = np.array(0, dtype=np.int64)


def _ln(sd, prefix, c, seed):
    sd[prefix + '.weight'] = _uniform(prefix + '.weight', seed, (c,), 0.5, 1.5)
    sd[prefix + '.bias'] = _normal(prefix + '.bias', seed, (c,), 0.1)


def _conv(sd, name, cout, cin, k, seed, gain=2.0):
    sd[name] = _normal(name, seed, (cout, cin, k, k), np.sqrt(gain / (cin * k * k)))


def _linear(sd, prefix, cout, cin, seed, bias=True, gain=1.0):
    sd[prefix + '.weight'] = _normal(prefix + '.weight', seed, (cout, cin), np.sqrt(gain / cin))
    if bias:
        sd[prefix + '.bias'] = _normal(prefix + '.bias', seed, (cout,), 0.1)


def make_state_dict(seed=0, depth=50, num_stages=4, d=256, ffn=2048, feat=64, roi=7, family='uniform'):
    """Synthetic weights with the reference detector's ``state_dict`` keys and shapes.

    ``family``: 'uniform' -- every tensor N(0, gain / fan_in), BN terms near one (rounds 1-3); 'trained' -- the statistics of a trained
    checkpoint where they matter to the arithmetic: He-init convs whose rows and BN sigma spread over three
    decades, BN gamma over two (folded weights from ~1e-3 to ~1 in one matrix), a small last BN of each block, and box-regression heads
    small enough that the refined boxes stay inside the frame (a trained detector's boxes sit on heads and faces).

    Returns ``dict[str, np.ndarray]`` (float32; ``num_batches_tracked`` int64).
    """
    assert family in ('uniform', 'trained'), family
    trained = family == 'trained'
    sd = {}
    # --- backbone (mmdet/models/backbones/resnet.py:369-391, arch table :361-367)
    _conv(sd, 'backbone.conv1.weight', 64, 3, 7, seed)
    _bn(sd, 'backbone.bn1', 64, seed)
    inplanes = 64
    for li, nblocks in enumerate(ARCH[depth]):
        planes = 64 * 2 ** li
        for bi in range(nblocks):
            p = f'backbone.layer{li + 1}.{bi}'
            _conv(sd, p + '.conv1.weight', planes, inplanes, 1, seed)
            _bn(sd, p + '.bn1', planes, seed)
            _conv(sd, p + '.conv2.weight', planes, planes, 3, seed)
            _bn(sd, p + '.bn2', planes, seed)
            _conv(sd, p + '.conv3.weight', planes * 4, planes, 1, seed, gain=1.0)
            _bn(sd, p + '.bn3', planes * 4, seed, gamma=(0.3, 0.7))
            if bi == 0:
                _conv(sd, p + '.downsample.0.weight', planes * 4, inplanes, 1, seed, gain=1.0)
                _bn(sd, p + '.downsample.1', planes * 4, seed)
            inplanes = planes * 4
    if trained:   # re-draw every BN of the backbone with the trained statistics (the conv rows are rescaled with them)
        _bn_trained(sd, 'backbone.bn1', 'backbone.conv1.weight', seed)
        for li, nblocks in enumerate(ARCH[depth]):
            for bi in range(nblocks):
                p = f'backbone.layer{li + 1}.{bi}'
                _bn_trained(sd, p + '.bn1', p + '.conv1.weight', seed)
                _bn_trained(sd, p + '.bn2', p + '.conv2.weight', seed)
                _bn_trained(sd, p + '.bn3', p + '.conv3.weight', seed, gamma=(0.01, 1.0))
                if bi == 0:
                    _bn_trained(sd, p + '.downsample.1', p + '.downsample.0.weight', seed)
    # --- FPN (mmdet/models/necks/fpn.py:110-126): conv + bias, no norm, no act
    for i, cin in enumerate((256, 512, 1024, 2048)):
        _conv(sd, f'neck.lateral_convs.{i}.conv.weight', d, cin, 1, seed, gain=1.0)
        sd[f'neck.lateral_convs.{i}.conv.bias'] = _normal(f'neck.lateral_convs.{i}.conv.bias', seed, (d,), 0.1)
        _conv(sd, f'neck.fpn_convs.{i}.conv.weight', d, d, 3, seed, gain=1.0)
        sd[f'neck.fpn_convs.{i}.conv.bias'] = _normal(f'neck.fpn_convs.{i}.conv.bias', seed, (d,), 0.1)
    # --- query embeddings (fixed_embedding_rpn_head.py:40-44): (cx, cy, w, h) normalised.
    # Chosen so the three queries start on P4 / P3 / P2.
    base = np.array([[0.5, 0.5, 0.9, 0.9], [0.5, 0.4, 0.45, 0.45], [0.5, 0.35, 0.2, 0.2]], dtype=np.float32)
    sd['rpn_head.init_proposal_bboxes.weight'] = base + _normal('rpn_head.init_proposal_bboxes.weight', seed, (3, 4), 0.02)
    sd['rpn_head.init_proposal_features.weight'] = _normal('rpn_head.init_proposal_features.weight', seed, (3, d), 1.0)
    for s in range(num_stages):
        p = f'roi_head.bbox_head.{s}'
        # dead BBoxHead leftovers (bbox_head.py:72-81) -- present in checkpoints, never used
        _linear(sd, p + '.fc_cls', 4, d * roi * roi, seed)
        _linear(sd, p + '.fc_reg', 4, d * roi * roi, seed)
        sd[p + '.attention.attn.in_proj_weight'] = _normal(p + '.attention.attn.in_proj_weight', seed, (3 * d, d), np.sqrt(1.0 / d))
        sd[p + '.attention.attn.in_proj_bias'] = _normal(p + '.attention.attn.in_proj_bias', seed, (3 * d,), 0.1)
        _linear(sd, p + '.attention.attn.out_proj', d, d, seed)
        _ln(sd, p + '.attention_norm', d, seed)
        q = p + '.instance_interactive_conv'
        _linear(sd, q + '.dynamic_layer', 2 * d * feat, d, seed)
        _ln(sd, q + '.norm_in', feat, seed)
        _ln(sd, q + '.norm_out', d, seed)
        _linear(sd, q + '.fc_layer', d, d * roi * roi, seed)
        _ln(sd, q + '.fc_norm', d, seed)
        _ln(sd, p + '.instance_interactive_conv_norm', d, seed)
        _linear(sd, p + '.ffn.layers.0.0', ffn, d, seed, gain=2.0)
        _linear(sd, p + '.ffn.layers.1', d, ffn, seed)
        _ln(sd, p + '.ffn_norm', d, seed)
        _linear(sd, p + '.cls_fcs.0', d, d, seed, bias=False, gain=2.0)
        _ln(sd, p + '.cls_fcs.1', d, seed)
        for j in range(3):
            _linear(sd, p + f'.reg_fcs.{3 * j}', d, d, seed, bias=False, gain=2.0)
            _ln(sd, p + f'.reg_fcs.{3 * j + 1}', d, seed)
        for clue in ('face', 'eyes', 'head'):
            _linear(sd, p + f'.{clue}_fc_cls', 1, d, seed)
            _linear(sd, p + f'.{clue}_fc_reg', 4, d, seed, gain=0.01 if trained else 0.25)
            if trained:
                sd[p + f'.{clue}_fc_reg.bias'] = (sd[p + f'.{clue}_fc_reg.bias'] * 0.1).astype(np.float32)
        g = f'roi_head.gaze_head.{s}'
        for clue in ('face', 'eyes', 'head'):
            for branch in (f'gaze_{clue}_fcs', f'gaze_{clue}_confidence'):
                for j in range(2):
                    _linear(sd, g + f'.{branch}.{3 * j}', d, d, seed, bias=False, gain=2.0)
                    _ln(sd, g + f'.{branch}.{3 * j + 1}', d, seed)
            _linear(sd, g + f'.fc_{clue}_confidence', 3, d, seed)
            _linear(sd, g + f'.fc_{clue}', 3, d, seed)
        _linear(sd, g + '.fc_gaze', 3, 9, seed)
    return sd


def make_clips(seed, num_clips, clip_length=7, height=224, width=224):
    """``[B*T, 3, H, W]`` float32 frames, N(0,1) in the normalised-pixel domain."""
    rs = np.random.RandomState(seed)
    return rs.standard_normal((num_clips * clip_length, 3, height, width)).astype(np.float32)


def make_img_metas(num_frames, img_shape=(224, 224, 3), pad_shape=None, scale_factor=(1., 1., 1., 1.)):
    """The per-frame meta dicts the reference passes beside the image tensor
    (tools/test_gaze360_gaze.py:96-101): img_shape, ori_shape, pad_shape, scale_factor."""
    pad_shape = pad_shape or img_shape
    return [dict(img_shape=tuple(img_shape), ori_shape=tuple(img_shape), pad_shape=tuple(pad_shape),
                 scale_factor=np.asarray(scale_factor, dtype=np.float32), flip=False,
                 filename=f'{i:05d}.png') for i in range(num_frames)]


def fake_clip_outputs(video_id, frames, call_index):
    """Seeded stand-in for the model's per-clip outputs, used to pin the windowing / overlap-merge logic
    (tests/test_harness.py, oracle/dev/make_harness_goldens.py): depends on the video, the frame numbers AND the
    call index, so the two predictions of an overlapped frame differ.  Returns torch tensors
    det_bboxes [T,3,5] (scores straddle the 0.5 person threshold), fused gaze [T,3], per-clue gazes [T,3,3]."""
    import torch
    T = len(frames)
    rs = np.random.RandomState((video_id * 7919 + call_index * 104729 + frames[0] * 31 + T) & 0x7FFFFFFF)
    xy = rs.uniform(5, 120, size=(T, 3, 2))
    wh = rs.uniform(10, 90, size=(T, 3, 2))
    score = rs.uniform(0.2, 1.0, size=(T, 3, 1))
    det = np.concatenate([xy, xy + wh, score], axis=-1).astype(np.float32)
    g = rs.standard_normal((T, 4, 3)).astype(np.float32)
    g /= np.linalg.norm(g, axis=-1, keepdims=True)
    return torch.from_numpy(det), torch.from_numpy(g[:, 0]), torch.from_numpy(g[:, 1:])


def fuzz_case(seed, index):
    """One addressable random input of tools/parity_fuzz.py / the parity tests: weight seed, batch, clip length, padded frame size
    (multiples of 32), per-frame img_shape inside it (40 % of the cases fill the frame), frames zero outside img_shape like the
    pipeline's padding.  -> dict(wseed, B, T, H, W, img_shape, img [N,3,H,W] f32, metas)."""
    rs = np.random.RandomState(1000003 * seed + index)
    wseed = int(rs.randint(0, 3))
    T = int(rs.choice([1, 2, 3, 5, 7, 7, 7, 9, 12]))
    B = int(rs.randint(1, 4))
    H, W = 32 * int(rs.randint(2, 11)), 32 * int(rs.randint(2, 11))
    full = rs.rand() < 0.4
    ih, iw = (H, W) if full else (int(rs.randint(H - 31, H + 1)), int(rs.randint(W - 31, W + 1)))
    img = make_clips(7000 + 13 * seed + index, B, T, H, W)
    if not full:
        img[:, :, ih:, :] = 0
        img[:, :, :, iw:] = 0
    return dict(wseed=wseed, B=B, T=T, H=H, W=W, img_shape=(ih, iw), full=full, img=img,
                metas=make_img_metas(B * T, (ih, iw, 3), pad_shape=(H, W, 3)))
