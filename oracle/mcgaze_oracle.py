"""ORACLE -- test infrastructure, NOT product code.

CPU fp32 restatement of the MCGaze per-clip forward path, written as
plain functions over a ``state_dict`` (``dict[str, Tensor]``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file;
the product package ``mcgaze_amd`` never does and fails loudly without its HIP library.

Each function cites the reference file:line (relative to the upstream repo root) whose
arithmetic it restates.  Arithmetic is float32 throughout, like the reference
(`fp16_enabled = False` everywhere).

Pinning: every function here is checked in the build container against the *imported*
reference Python (``oracle/dev/make_goldens.py`` -> ``tests/golden/*.npz``; the generator
imports ``/root/reference/mmdet`` through a stand-in for the absent ``mmcv`` package) and
against the one known-answer vector the reference's own tests hold for this path
(``tests/test_utils/test_coder.py:27-76``, the ``delta2bbox`` KAT).
PARITY UNPINNED at one boundary: ``mmcv.ops.RoIAlign`` (mmcv-full 1.4.8) is a third-party
native op whose source is not in the reference tree and the reference has no test that
fixes its values; ``roi_align`` below restates the published mmcv/Detectron2 definition
(``aligned=True``) and is pinned only by hand-computed cases (tests/test_oracle.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

ARCH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
CLUES = ('face', 'eyes', 'head')  # token order: gaze_stqi_head.py:191-201, gaze_head.py:155-157


def as_torch(sd):
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}


# --------------------------------------------------------------------------- trunk
def _bn(sd, p, x):
    # BN in eval mode (norm_eval=True, resnet.py:648-658); eps = torch default 1e-5
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, 1e-5)


def bottleneck(sd, p, x, stride):
    """resnet.py:263-302, style='pytorch' (stride on the 3x3, :154-159)."""
    out = F.relu(_bn(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'])))
    out = F.relu(_bn(sd, p + '.bn2', F.conv2d(out, sd[p + '.conv2.weight'], stride=stride, padding=1)))
    out = _bn(sd, p + '.bn3', F.conv2d(out, sd[p + '.conv3.weight']))
    if (p + '.downsample.0.weight') in sd:  # res_layer.py:51-61
        x = _bn(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride))
    return F.relu(out + x)


def resnet(sd, img, depth=50):
    """resnet.py:631-646 -> (C2, C3, C4, C5)."""
    x = F.relu(_bn(sd, 'backbone.bn1', F.conv2d(img, sd['backbone.conv1.weight'], stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)  # resnet.py:611
    outs = []
    for li, nblocks in enumerate(ARCH[depth]):
        for bi in range(nblocks):
            x = bottleneck(sd, f'backbone.layer{li + 1}.{bi}', x, stride=(2 if (bi == 0 and li > 0) else 1))
        outs.append(x)
    return outs


def fpn(sd, feats):
    """fpn.py:157-180 (start_level=0, num_outs=4, no extra convs, nearest top-down)."""
    lat = [F.conv2d(f, sd[f'neck.lateral_convs.{i}.conv.weight'], sd[f'neck.lateral_convs.{i}.conv.bias'])
           for i, f in enumerate(feats)]
    for i in range(len(lat) - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    return [F.conv2d(l, sd[f'neck.fpn_convs.{i}.conv.weight'], sd[f'neck.fpn_convs.{i}.conv.bias'], padding=1)
            for i, l in enumerate(lat)]


# --------------------------------------------------------------------------- queries / boxes
def init_proposals(sd, img_metas):
    """fixed_embedding_rpn_head.py:76-94."""
    e = sd['rpn_head.init_proposal_bboxes.weight']
    cx, cy, w, h = e[:, 0:1], e[:, 1:2], e[:, 2:3], e[:, 3:4]
    xyxy = torch.cat([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)  # transforms.py:245-256
    whwh = torch.tensor([[m['img_shape'][1], m['img_shape'][0]] * 2 for m in img_metas], dtype=torch.float32)
    boxes = xyxy[None] * whwh[:, None, :]
    feats = sd['rpn_head.init_proposal_features.weight'][None].expand(len(img_metas), -1, -1).clone()
    return boxes, feats


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000, clip_border=True):
    """delta_xywh_bbox_coder.py:224-260 (class-agnostic, no ctr clamp)."""
    means = deltas.new_tensor(means).view(1, -1)
    stds = deltas.new_tensor(stds).view(1, -1)
    d = deltas * stds + means
    pxy = (rois[:, :2] + rois[:, 2:]) * 0.5
    pwh = rois[:, 2:] - rois[:, :2]
    max_ratio = np.abs(np.log(wh_ratio_clip))
    dwh = d[:, 2:].clamp(min=-max_ratio, max=max_ratio)
    gxy = pxy + pwh * d[:, :2]
    gwh = pwh * dwh.exp()
    out = torch.cat([gxy - gwh * 0.5, gxy + gwh * 0.5], dim=-1)
    if clip_border and max_shape is not None:
        out[..., 0::2].clamp_(min=0, max=max_shape[1])
        out[..., 1::2].clamp_(min=0, max=max_shape[0])
    return out


def map_roi_levels(boxes, num_levels=4, finest_scale=56):
    """single_level_roi_extractor.py:36-55; boxes [R,4] xyxy."""
    scale = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lvl = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lvl.clamp(min=0, max=num_levels - 1).long()


def _bilinear_axis(c, L):
    """Per-axis part of mmcv's bilinear_interpolate (roi_align kernel, aligned=True):
    out-of-range flag for c < -1 or c > L; clamp at 0; at the far edge low=high=L-1."""
    invalid = (c < -1.0) | (c > L)
    c = c.clamp(min=0)
    lo = c.floor().long()
    top = lo >= L - 1
    lo = torch.where(top, torch.full_like(lo, L - 1), lo)
    hi = torch.where(top, lo, lo + 1)
    c = torch.where(top, lo.to(c.dtype), c)
    l = c - lo.to(c.dtype)
    return lo, hi, l, 1 - l, invalid


def roi_align(feat, rois, spatial_scale, out_size=7, sampling_ratio=2):
    """mmcv.ops.RoIAlign(output_size, spatial_scale, sampling_ratio, 'avg', aligned=True)
    as called from base_roi_extractor.py:54-60 / single_level_roi_extractor.py:96.
    feat [N,C,H,W]; rois [K,5] (frame, x1,y1,x2,y2) in image pixels -> [K,C,7,7]."""
    N, C, H, W = feat.shape
    K, s, P = rois.shape[0], sampling_ratio, out_size
    b = rois[:, 0].long()
    x1, y1 = rois[:, 1] * spatial_scale - 0.5, rois[:, 2] * spatial_scale - 0.5
    x2, y2 = rois[:, 3] * spatial_scale - 0.5, rois[:, 4] * spatial_scale - 0.5
    bw, bh = (x2 - x1) / P, (y2 - y1) / P
    g = (torch.arange(P * s, dtype=feat.dtype) + 0.5) / s  # sample positions in units of bins
    ys = y1[:, None] + g[None] * bh[:, None]
    xs = x1[:, None] + g[None] * bw[:, None]
    ylo, yhi, ly, hy, yinv = _bilinear_axis(ys, H)
    xlo, xhi, lx, hx, xinv = _bilinear_axis(xs, W)
    ff = feat.reshape(N, C, H * W)
    bi, ci = b[:, None, None], torch.arange(C)[None, :, None]

    def gather(yi, xi):   # the K x C x (P s)^2 taps straight out of feat (a per-RoI copy feat[b] would move K whole maps: 3.2 MB each on P2)
        idx = (yi[:, :, None] * W + xi[:, None, :]).reshape(K, 1, -1)
        return ff[bi, ci, idx].reshape(K, C, P * s, P * s)

    def w(a, bb):
        return (a[:, :, None] * bb[:, None, :])[:, None]

    v = gather(ylo, xlo) * w(hy, hx) + gather(ylo, xhi) * w(hy, lx) + gather(yhi, xlo) * w(ly, hx) + gather(yhi, xhi) * w(ly, lx)
    v = torch.where((yinv[:, :, None] | xinv[:, None, :])[:, None], torch.zeros_like(v), v)
    return v.reshape(K, C, P, s, P, s).mean(dim=(3, 5))


def roi_align_scalar(feat, rois, spatial_scale, out_size=7, sampling_ratio=2):
    """Loop form of the same definition (numpy, one sample at a time) -- the readable
    statement of the kernel; used to pin ``roi_align`` above on small cases."""
    feat = np.asarray(feat, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    N, C, H, W = feat.shape
    P, s = out_size, sampling_ratio
    out = np.zeros((rois.shape[0], C, P, P), dtype=np.float32)
    f32 = np.float32
    for k, r in enumerate(rois):
        b = int(r[0])
        x1, y1 = f32(r[1] * f32(spatial_scale) - f32(0.5)), f32(r[2] * f32(spatial_scale) - f32(0.5))
        x2, y2 = f32(r[3] * f32(spatial_scale) - f32(0.5)), f32(r[4] * f32(spatial_scale) - f32(0.5))
        bw, bh = f32((x2 - x1) / f32(P)), f32((y2 - y1) / f32(P))
        for ph in range(P):
            for pw in range(P):
                acc = np.zeros(C, dtype=np.float32)
                for iy in range(s):
                    y = f32(y1 + f32(ph) * bh + f32(iy + 0.5) * bh / f32(s))
                    for ix in range(s):
                        x = f32(x1 + f32(pw) * bw + f32(ix + 0.5) * bw / f32(s))
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        yy, xx = max(y, f32(0)), max(x, f32(0))
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yh = yl = H - 1
                            yy = f32(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xh = xl = W - 1
                            xx = f32(xl)
                        else:
                            xh = xl + 1
                        ly, lx = f32(yy - yl), f32(xx - xl)
                        hy, hx = f32(1) - ly, f32(1) - lx
                        acc += (hy * hx) * feat[b, :, yl, xl] + (hy * lx) * feat[b, :, yl, xh] \
                            + (ly * hx) * feat[b, :, yh, xl] + (ly * lx) * feat[b, :, yh, xh]
                out[k, :, ph, pw] = acc / f32(s * s)
    return out


def roi_extract(fpn_feats, boxes, strides=(4, 8, 16, 32)):
    """single_level_roi_extractor.py:57-115. boxes [N,P,4] -> roi feats [N*P,256,7,7]."""
    N, P = boxes.shape[:2]
    frame = torch.arange(N, dtype=boxes.dtype)[:, None].expand(N, P).reshape(-1, 1)
    rois = torch.cat([frame, boxes.reshape(-1, 4)], dim=1)  # bbox2roi, transforms.py:75-94
    lvls = map_roi_levels(rois[:, 1:])
    out = fpn_feats[0].new_zeros(rois.shape[0], fpn_feats[0].shape[1], 7, 7)
    for i, f in enumerate(fpn_feats):
        inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel():
            out[inds] = roi_align(f, rois[inds], 1.0 / strides[i])
    return out


# --------------------------------------------------------------------------- decoder stage
def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def mha_self(sd, p, x, num_heads=8):
    """mmcv MultiheadAttention wrapper = identity + nn.MultiheadAttention(x,x,x)[0]
    (sequence-first; gaze_stqi_head.py:51,151,162).  x [L, Bt, d]."""
    L, Bt, d = x.shape
    hd = d // num_heads
    qkv = F.linear(x, sd[p + '.attn.in_proj_weight'], sd[p + '.attn.in_proj_bias'])
    q, k, v = qkv.split(d, dim=-1)

    def heads(t):  # [L,Bt,d] -> [Bt*h, L, hd]
        return t.reshape(L, Bt * num_heads, hd).transpose(0, 1)

    q, k, v = heads(q) * (1.0 / math.sqrt(hd)), heads(k), heads(v)
    a = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    o = torch.bmm(a, v).transpose(0, 1).reshape(L, Bt, d)
    o = F.linear(o, sd[p + '.attn.out_proj.weight'], sd[p + '.attn.out_proj.bias'])
    return x + o


def dynamic_conv(sd, p, x, roi, feat=64):
    """transformer.py:1116-1164. x [R,d]; roi [R,d,7,7] -> [R,d]."""
    R, d = x.shape
    f = roi.flatten(2).permute(0, 2, 1)  # [R,49,d]
    theta = F.linear(x, sd[p + '.dynamic_layer.weight'], sd[p + '.dynamic_layer.bias'])
    w_in = theta[:, :d * feat].view(R, d, feat)
    w_out = theta[:, -d * feat:].view(R, feat, d)
    f = F.relu(_ln(sd, p + '.norm_in', torch.bmm(f, w_in)))
    f = F.relu(_ln(sd, p + '.norm_out', torch.bmm(f, w_out)))
    f = F.linear(f.flatten(1), sd[p + '.fc_layer.weight'], sd[p + '.fc_layer.bias'])
    return F.relu(_ln(sd, p + '.fc_norm', f))


def stqi_stage(sd, s, roi_feat, obj, clip_length, return_intermediates=False):
    """GazeSTQIHead.forward, gaze_stqi_head.py:119-202.
    roi_feat [N*3,d,7,7]; obj [N,3,d] -> cls [N,3,1], delta [N,3,4], obj' [N,3,d]."""
    p = f'roi_head.bbox_head.{s}'
    N, P, d = obj.shape
    T = clip_length
    x = obj.permute(1, 0, 2)  # spatial: seq = 3 clues, batch = N frames (:148-151)
    x = _ln(sd, p + '.attention_norm', mha_self(sd, p + '.attention', x)).permute(1, 0, 2)
    sp = x
    x = x.reshape(N // T, T, P, d).permute(1, 0, 2, 3).reshape(T, N * P // T, d)  # temporal (:156-166)
    x = _ln(sd, p + '.attention_norm', mha_self(sd, p + '.attention', x))
    x = x.reshape(T, N // T, P, d).permute(1, 0, 2, 3).reshape(N, P, d)
    attn = x
    x = x.reshape(-1, d)
    x = _ln(sd, p + '.instance_interactive_conv_norm', x + dynamic_conv(sd, p + '.instance_interactive_conv', x, roi_feat))
    iic = x
    h = F.linear(F.relu(F.linear(x, sd[p + '.ffn.layers.0.0.weight'], sd[p + '.ffn.layers.0.0.bias'])),
                 sd[p + '.ffn.layers.1.weight'], sd[p + '.ffn.layers.1.bias'])
    x = _ln(sd, p + '.ffn_norm', x + h).view(N, P, d)  # mmcv FFN: add_identity (:179-180)
    cls_f = F.relu(_ln(sd, p + '.cls_fcs.1', F.linear(x, sd[p + '.cls_fcs.0.weight'])))
    reg_f = x
    for j in range(3):
        reg_f = F.relu(_ln(sd, p + f'.reg_fcs.{3 * j + 1}', F.linear(reg_f, sd[p + f'.reg_fcs.{3 * j}.weight'])))
    cls = torch.stack([F.linear(cls_f[:, c], sd[p + f'.{n}_fc_cls.weight'], sd[p + f'.{n}_fc_cls.bias'])
                       for c, n in enumerate(CLUES)], dim=1)
    delta = torch.stack([F.linear(reg_f[:, c], sd[p + f'.{n}_fc_reg.weight'], sd[p + f'.{n}_fc_reg.bias'])
                         for c, n in enumerate(CLUES)], dim=1)
    if return_intermediates:
        return cls, delta, x, dict(spatial=sp, attn=attn, iic=iic)
    return cls, delta, x


def gaze_head(sd, s, obj):
    """GazeHead.forward, gaze_head.py:138-202. obj [N,3,d] -> 4 x [N,3] unit vectors."""
    g = f'roi_head.gaze_head.{s}'

    def mlp(branch, x):
        for j in range(2):
            x = F.relu(_ln(sd, g + f'.{branch}.{3 * j + 1}', F.linear(x, sd[g + f'.{branch}.{3 * j}.weight'])))
        return x

    gz, conf = [], []
    for c, n in enumerate(CLUES):
        f = obj[:, c, :]
        gz.append(F.linear(mlp(f'gaze_{n}_fcs', f), sd[g + f'.fc_{n}.weight'], sd[g + f'.fc_{n}.bias']))
        conf.append(F.linear(mlp(f'gaze_{n}_confidence', f), sd[g + f'.fc_{n}_confidence.weight'], sd[g + f'.fc_{n}_confidence.bias']))
    fused = F.linear(torch.cat([c * z for c, z in zip(conf, gz)], dim=1), sd[g + '.fc_gaze.weight'], sd[g + '.fc_gaze.bias'])
    unit = lambda t: t / torch.norm(t, dim=-1, keepdim=True)  # no epsilon (:197-200)
    return dict(gaze_score=unit(fused), face_gaze_score=unit(gz[0]), eyes_gaze_score=unit(gz[1]), head_gaze_score=unit(gz[2]))


# --------------------------------------------------------------------------- whole path
def decoder(sd, fpn_feats, img_metas, clip_length, num_stages=4, stds=(0.5, 0.5, 1., 1.), collect=None):
    """MultiClueGazeROIHead.simple_test loop (multiclue_gaze_roi_head.py:337-379) with the
    batched semantics of forward_train (:228-229): ``clip_length=T``, N = B*T frames."""
    boxes, obj = init_proposals(sd, img_metas)
    N = boxes.shape[0]
    cls = None
    for s in range(num_stages):
        roi = roi_extract(fpn_feats, boxes)
        cls, delta, obj = stqi_stage(sd, s, roi, obj, clip_length)
        # refine_bboxes -> regress_by_class -> DeltaXYWHBBoxCoder.decode, clip_border=False
        boxes = delta2bbox(boxes.reshape(-1, 4), delta.reshape(-1, 4), stds=stds, clip_border=False).reshape(N, -1, 4)
        if collect is not None:
            collect.append(dict(obj=obj.clone(), boxes=boxes.clone(), cls=cls.clone()))
    gaze = gaze_head(sd, num_stages - 1, obj)  # last stage's head on last stage's obj_feat (:367,:378)
    return boxes, cls.sigmoid(), gaze


def forward(sd, img, img_metas, clip_length, rescale=False, depth=50, collect=None):
    """MultiClueGaze.simple_test (multiclue_gaze.py:105-131), batched over B = N/clip_length clips.
    Returns (det_bboxes [N,3,5], gaze dict of 4 x [N,3])."""
    sd = as_torch(sd)
    img = torch.as_tensor(img, dtype=torch.float32)
    with torch.no_grad():
        feats = fpn(sd, resnet(sd, img, depth))
        boxes, scores, gaze = decoder(sd, feats, img_metas, clip_length, collect=collect)
        if rescale:  # multiclue_gaze_roi_head.py:360-363
            sf = torch.tensor(np.stack([np.asarray(m['scale_factor'], dtype=np.float32) for m in img_metas]))
            boxes = boxes / sf[:, None, :]
        return torch.cat([boxes, scores], dim=-1), gaze


def wrap_yaw(d):
    """A (yaw, pitch) DIFFERENCE with its yaw component taken modulo 2 pi into (-pi, pi]: yaw = atan2(x, -z) has its branch cut where
    the gaze points straight back, and two vectors 1e-5 rad apart on either side of it differ by 2 pi in the raw subtraction
    (tools/parity_fuzz.py, seed 11 case 891)."""
    d = d.clone()
    d[..., 0] = torch.remainder(d[..., 0] + math.pi, 2 * math.pi) - math.pi
    return d


def yaw_pitch_diff(a, b):
    """|(yaw, pitch)(a) - (yaw, pitch)(b)| per component, yaw wrapped: the quantity north_star bounds by 1e-3."""
    return wrap_yaw(yaw_pitch(a) - yaw_pitch(b)).abs()


def yaw_pitch(g):
    """(yaw, pitch) = (atan2(x, -z), asin(y)) -- the comparison domain north_star names
    (yaw definition: tools/calculate_mae_gaze360.py:60-74)."""
    g = torch.as_tensor(g, dtype=torch.float32)
    return torch.stack([torch.atan2(g[..., 0], -g[..., 2]), torch.asin(g[..., 1].clamp(-1, 1))], dim=-1)
