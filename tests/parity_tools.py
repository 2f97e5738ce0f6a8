"""Shared by tests/test_gpu_forward.py and tools/parity_fuzz.py: how an engine's deviation from the CPU oracle is taken apart.

The model is not a continuous function of its inputs: RoIAlign drops a sample whose coordinate leaves [-1, L]
(mmcv roi_align kernel; oracle/mcgaze_oracle.py::_bilinear_axis) and the RoI -> pyramid-level map is a floor(log2())
(single_level_roi_extractor.py:51-54); and (yaw, pitch) is singular where the gaze points straight up or down.  So an
end-to-end bound on (yaw, pitch) cannot hold for EVERY input of ANY implementation whose rounding differs from the reference's;
what can is (1) every stage's arithmetic on identical inputs ("teacher-forced") and (2) the end-to-end bound wherever the chain
crosses no discontinuity.  These helpers measure both."""
import numpy as np
import torch

from mcgaze_amd import engine as E
from oracle import mcgaze_oracle as orc


def sample_validity(boxes, levels, hs, ws, strides=(4, 8, 16, 32)):
    """Per box the 28 'sample inside [-1, L]' flags of RoIAlign (14 per axis)."""
    out = []
    for (x1, y1, x2, y2), lv in zip(torch.as_tensor(boxes).reshape(-1, 4).tolist(), torch.as_tensor(levels).tolist()):
        sc = 1.0 / strides[lv]
        flags = []
        for a0, a1, L in ((y1, y2, hs[lv]), (x1, x2, ws[lv])):
            start, end = a0 * sc - 0.5, a1 * sc - 0.5
            b = (end - start) / 7
            cs = [start + (i // 2) * b + ((i % 2) + 0.5) * b / 2 for i in range(14)]
            flags += [not (c < -1.0 or c > L) for c in cs]
        out.append(flags)
    return np.array(out)


def stage_report(eng, prec, sd, img, metas, T, stages):
    """stages: the oracle's per-stage outputs (orc.forward(..., collect=[])).  Returns dict(
    teacher_forced = [obj error of scale per stage, engine stage fed with the ORACLE's inputs],
    free_running   = [(level flips, boxes with a sample-validity flip, obj error of scale, max box |d| px) per stage],
    discontinuity  = whether the engine's own chain crossed a level / validity boundary the oracle's did not,
    crossed_boxes  = bool [R]: the boxes (row = frame * 3 + clue) that did, in any stage)."""
    split = prec == 'f16x3'
    pyr = eng.backbone_fpn(torch.from_numpy(np.ascontiguousarray(img)).to(eng.device))
    hs, ws = [p.shape[1] for p in pyr], [p.shape[2] for p in pyr]
    boxes0, obj0 = orc.init_proposals(orc.as_torch(sd), metas)
    dt, dev = pyr[0].dtype, eng.device
    tf, fr = [], []
    bin_, oin = boxes0, obj0
    for s in range(4):
        roi, _ = E.roi_align(pyr, bin_.to(dev).contiguous())
        o, _, _ = E.stage_forward(eng.weights.stages[s], roi, oin.to(dt).to(dev).contiguous(), bin_.to(dev).contiguous(), T, split=split)
        want = stages[s]['obj']
        tf.append(float((o.float().cpu() - want).abs().max() / want.abs().max()))
        bin_, oin = stages[s]['boxes'], stages[s]['obj']
    b, o, ref_in = boxes0.to(dev).contiguous(), obj0.to(dt).to(dev).contiguous(), boxes0
    crossed = np.zeros(boxes0.reshape(-1, 4).shape[0], dtype=bool)   # per box: a level or sample-validity flip in ANY stage
    for s in range(4):
        roi, lv = E.roi_align(pyr, b)
        ref_lv = orc.map_roi_levels(ref_in.reshape(-1, 4))
        flips = int((lv.cpu().long() != ref_lv).sum())
        same_lv = (lv.cpu().long() == ref_lv).numpy()
        vf = (sample_validity(b.cpu(), lv.cpu(), hs, ws) != sample_validity(ref_in, ref_lv, hs, ws)).any(axis=1)
        vflips = int((vf & same_lv).sum())
        crossed |= (~same_lv) | vf
        o, b, _ = E.stage_forward(eng.weights.stages[s], roi, o, b, T, split=split)
        want = stages[s]['obj']
        fr.append((flips, vflips, float((o.float().cpu() - want).abs().max() / want.abs().max()), float((b.cpu() - stages[s]['boxes']).abs().max())))
        ref_in = stages[s]['boxes']
    return dict(teacher_forced=tf, free_running=fr, discontinuity=any(a or v for a, v, _, _ in fr), crossed_boxes=crossed)


def describe(rep):
    return ('teacher-forced stage errors (of scale) ' + ', '.join(f'{e:.1e}' for e in rep['teacher_forced']) +
            '; free-running per stage (level flips, boxes with a sample-validity flip, obj error of scale, max box |d| px): ' +
            ', '.join(f'({a}, {v}, {e:.1e}, {d:.2g})' for a, v, e, d in rep['free_running']))
