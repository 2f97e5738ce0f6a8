"""Gaze360 / L2CS mean-angular-error metric, restated from
tools/calculate_mae_gaze360.py:16-29,60-94,110-188 and tools/calculate_mae_l2cs.py (annotation index
``anno_id*3`` :110, front-20 also needs |pitch| <= 20 :132-139).  Behaviours kept on purpose: predictions are
temporally smoothed (alpha=0.6, 3-tap interior, 2-tap ends, re-normalised), targets are normalised, predictions
are NOT re-normalised inside the angular error and ``acos`` is applied unclamped; per-video errors are averaged
weighted by frame count; the README's 10.74 is the "front 90" (|yaw| <= 90 deg) line."""
import math

import torch


def smooth_filter(x, alpha=0.6):
    if x.size(0) >= 2:
        out = alpha * x
        out[0] += (1 - alpha) * x[1]
        out[-1] += (1 - alpha) * x[-2]
        out[1:-1] += (1 - alpha) * (x[0:-2] + x[2:]) / 2
        return out / torch.norm(out, dim=1).unsqueeze(1)
    return x


def vector_to_yaw_pitch(v):
    v = torch.reshape(v, (-1, 3))
    v = v / torch.norm(v, dim=1).reshape(-1, 1)
    return torch.stack([torch.atan2(v[:, 0], -v[:, 2]), torch.asin(v[:, 1])], dim=1)


def compute_yaw_angular(target):
    return 180 * torch.abs(vector_to_yaw_pitch(target)[:, 0]) / math.pi


def compute_pitch_angular(target):
    return 180 * torch.abs(vector_to_yaw_pitch(target)[:, 1]) / math.pi


def compute_angular_error(pred, target):
    target = target / torch.norm(target, dim=1).unsqueeze(1)
    dot = torch.bmm(target.view(-1, 1, 3), pred.view(-1, 3, 1)).view(-1)
    return 180 * torch.mean(torch.acos(dot)) / math.pi


def gaze_error(eval_data, anno_data, gaze_name='fusion_gazes', setting='gaze360', verbose=True):
    """-> dict(mae_360, mae_front_90, mae_front_20); prints the reference's three lines when verbose."""
    assert setting in ('gaze360', 'l2cs')
    tot = dict(n360=0, nf=0, n20=0, e360=0.0, ef=0.0, e20=0.0)
    for anno_id, video in enumerate(eval_data):
        pred = torch.tensor(video[gaze_name])
        gt = torch.tensor(anno_data['annotations'][anno_id * 3 if setting == 'l2cs' else anno_id]['gaze'])
        assert len(gt) == len(pred)
        pred = smooth_filter(pred)
        yaw = compute_yaw_angular(gt)
        front = yaw <= 90
        front20 = (yaw <= 20) & (compute_pitch_angular(gt) <= 20) if setting == 'l2cs' else yaw <= 20
        L = len(pred)
        tot['n360'] += L
        tot['e360'] += compute_angular_error(pred, gt) * L
        if int(front.sum()) > 0:
            tot['nf'] += int(front.sum())
            tot['ef'] += compute_angular_error(pred[front], gt[front]) * int(front.sum())
        if int(front20.sum()) > 0:
            tot['n20'] += int(front20.sum())
            tot['e20'] += compute_angular_error(pred[front20], gt[front20]) * int(front20.sum())
    out = dict(mae_360=float(tot['e360'] / tot['n360']), mae_front_90=float(tot['ef'] / tot['nf']) if tot['nf'] else float('nan'),
               mae_front_20=float(tot['e20'] / tot['n20']) if tot['n20'] else float('nan'))
    if verbose:
        print('%s mean angular error 360: %.2f' % (gaze_name, out['mae_360']))
        print('%s mean angular front 90: %.2f' % (gaze_name, out['mae_front_90']))
        print('%s mean angular front 20: %.2f\n' % (gaze_name, out['mae_front_20']))
    return out
