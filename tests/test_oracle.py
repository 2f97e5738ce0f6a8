"""The oracle (oracle/mcgaze_oracle.py) against the committed golden vectors, which were
produced by the reference's own Python (oracle/dev/make_goldens.py), and against the one
known-answer vector the reference's tests hold for this path (test_coder.py:27-76)."""
import os

import numpy as np
import pytest
import torch

from mcgaze_amd import synth
from oracle import mcgaze_oracle as orc

CASES = ['clip224', 'clip_nonsquare', 'batch2', 'clip_t5', 'trained_clip224', 'trained_nonsquare_b2']   # trained_*: synth's 'trained' weight family


def load_case(golden_dir, name):
    g = dict(np.load(os.path.join(golden_dir, name + '.npz')))
    B, T = int(g['B']), int(g['T'])
    ishape, pshape = tuple(int(v) for v in g['img_shape']), tuple(int(v) for v in g['pad_shape'])
    img = synth.make_clips(int(g['img_seed']), B, T, pshape[0], pshape[1])
    if ishape != pshape:
        img[:, :, ishape[0]:, :] = 0
        img[:, :, :, ishape[1]:] = 0
    metas = synth.make_img_metas(B * T, ishape, pshape, tuple(float(v) for v in g['scale_factor']))
    return g, img, metas, T


@pytest.fixture(scope='module')
def weights():
    return orc.as_torch(synth.make_state_dict(0))


@pytest.fixture(scope='module')
def weights_trained():
    return orc.as_torch(synth.make_state_dict(0, family='trained'))


@pytest.mark.parametrize('name', CASES)
def test_forward_matches_reference_golden(golden_dir, weights, weights_trained, name):
    g, img, metas, T = load_case(golden_dir, name)
    if str(g.get('weight_family', 'uniform')) == 'trained':
        weights = weights_trained
    col = []
    det, gaze = orc.forward(weights, img, metas, T, rescale=bool(g['rescale']), collect=col)
    for k in ('gaze_score', 'face_gaze_score', 'eyes_gaze_score', 'head_gaze_score'):
        np.testing.assert_allclose(gaze[k].numpy(), g[k], atol=2e-5, rtol=0)
    # north_star tolerance domain: (yaw, pitch) within 1e-3 -- the oracle sits at ~1e-6
    assert orc.yaw_pitch_diff(gaze['gaze_score'], g['gaze_score']).max() < 1e-4
    np.testing.assert_allclose(det.numpy(), g['det_bboxes'], atol=5e-3, rtol=1e-5)
    for s, c in enumerate(col):
        np.testing.assert_allclose(c['obj'].numpy(), g['stage_obj'][s], atol=1e-4, rtol=0)
        np.testing.assert_allclose(c['boxes'].numpy(), g['stage_boxes'][s], atol=5e-3, rtol=1e-5)
        np.testing.assert_allclose(c['cls'].numpy(), g['stage_cls'][s], atol=1e-4, rtol=0)


def test_trunk_matches_reference_golden(golden_dir, weights):
    g, img, metas, T = load_case(golden_dir, 'clip_nonsquare')
    with torch.no_grad():
        feats = orc.fpn(weights, orc.resnet(weights, torch.from_numpy(img)))
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g[f'fpn{i}_shape'])
        np.testing.assert_allclose(f.reshape(-1)[g[f'fpn{i}_idx']].numpy(), g[f'fpn{i}_val'], atol=2e-4, rtol=1e-5)
        assert abs(f.abs().mean().item() - float(g[f'fpn{i}_absmean'])) < 1e-4
    boxes, _ = orc.init_proposals(weights, metas)
    np.testing.assert_allclose(boxes.numpy(), g['init_boxes'], atol=1e-4)
    roi = orc.roi_extract(feats, boxes)
    np.testing.assert_allclose(roi.reshape(-1)[g['stage0_roi_idx']].numpy(), g['stage0_roi_val'], atol=2e-4, rtol=1e-5)


def test_coder_known_answer(golden_dir):
    """tests/test_utils/test_coder.py:27-76 / delta_xywh_bbox_coder.py:210-222."""
    k = np.load(os.path.join(golden_dir, 'coder_kat.npz'))
    out = orc.delta2bbox(torch.from_numpy(k['rois']), torch.from_numpy(k['deltas']), max_shape=(32, 32))
    expected = torch.tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                             [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]])
    assert torch.allclose(out, expected, atol=1e-4)
    np.testing.assert_allclose(out.numpy(), k['decoded'], atol=1e-6)


def test_roi_level_routing():
    """finest_scale=56: <112 -> P2, <224 -> P3, <448 -> P4, else P5; a 224-px box maps to
    level 2 (single_level_roi_extractor.py:36-55)."""
    b = torch.tensor([[0, 0, 111.9, 111.9], [0, 0, 112, 112], [0, 0, 224, 224], [0, 0, 448, 448], [0, 0, 1, 1]])
    assert orc.map_roi_levels(b).tolist() == [0, 1, 2, 3, 0]


def test_roi_align_hand_cases():
    """mmcv RoIAlign(aligned=True) definition pinned by hand: constant map -> constant;
    linear ramp -> bin-centre values; fully outside -> zeros; the scalar and the
    vectorised statements agree on random boxes including ones that leave the map."""
    H = W = 8
    const = torch.full((1, 2, H, W), 3.5)
    rois = torch.tensor([[0, 1.0, 1.0, 6.0, 6.0]])
    assert torch.allclose(orc.roi_align(const, rois, 1.0), torch.full((1, 2, 7, 7), 3.5))
    ramp = torch.arange(W, dtype=torch.float32)[None, None, None, :].expand(1, 1, H, W).contiguous()
    # roi x in [1.5, 5.0] (aligned -> [1.0, 4.5]), bin width 0.5, bin centres 1.25 + 0.5*i
    out = orc.roi_align(ramp, torch.tensor([[0, 1.5, 1.5, 5.0, 5.0]]), 1.0)
    assert torch.allclose(out[0, 0, 0], 1.25 + 0.5 * torch.arange(7.0), atol=1e-6)
    far = orc.roi_align(ramp, torch.tensor([[0, 20.0, 20.0, 30.0, 30.0]]), 1.0)
    assert float(far.abs().max()) == 0.0
    rs = np.random.RandomState(3)
    feat = torch.from_numpy(rs.standard_normal((2, 3, 6, 9)).astype(np.float32))
    xy = rs.uniform(-6, 40, size=(12, 2)).astype(np.float32)
    wh = rs.uniform(1, 30, size=(12, 2)).astype(np.float32)
    rois = np.concatenate([rs.randint(0, 2, size=(12, 1)).astype(np.float32), xy, xy + wh], axis=1)
    a = orc.roi_align(feat, torch.from_numpy(rois), 0.25).numpy()
    b = orc.roi_align_scalar(feat.numpy(), rois, 0.25)
    np.testing.assert_allclose(a, b, atol=2e-6)


def test_batched_semantics_equal_per_clip(golden_dir, weights):
    """N = B*T frames with clip_length = T equals B separate clips."""
    g, img, metas, T = load_case(golden_dir, 'batch2')
    _, both = orc.forward(weights, img, metas, T)
    _, first = orc.forward(weights, img[:T], metas[:T], T)
    _, second = orc.forward(weights, img[T:], metas[T:], T)
    sep = torch.cat([first['gaze_score'], second['gaze_score']])
    assert (both['gaze_score'] - sep).abs().max() < 2e-5


def test_roi_align_independent_pins():
    """RoIAlign against pins derived from the PUBLISHED mmcv kernel, not from oracle code (tests/roi_align_pins.py): closed-form
    values on an affine map for every enumerated edge case -- rows in (-1, 0), y == -1, y == H, the far-edge collapse, zero-area and
    inverted boxes, samples outside on either side -- plus hand-worked numbers on a non-affine map.  Both the vectorised oracle
    (used by the end-to-end goldens) and the scalar statement must reproduce them."""
    from tests import roi_align_pins as P
    H = W = 16
    feat = P.affine_map(H, W, base=7.0, channels=2)
    for name, box in P.EDGE_BOXES:
        want = P.expected_affine(box, 4, H, W, base=7.0, channels=2)
        rois = np.array([[0, *box]], dtype=np.float32)
        got_v = orc.roi_align(torch.from_numpy(feat), torch.from_numpy(rois), 0.25).numpy()[0]
        got_s = orc.roi_align_scalar(feat, rois, 0.25)[0]
        np.testing.assert_allclose(got_v, want, rtol=0, atol=2e-4, err_msg=name)
        np.testing.assert_allclose(got_s, want, rtol=0, atol=2e-4, err_msg=name)
    # the cases exercise what they claim to
    assert P.expected_affine(P.EDGE_BOXES[3][1], 4, H, W)[0, 0].max() == 0.0 and P.expected_affine(P.EDGE_BOXES[3][1], 4, H, W)[0, 6].min() > 0
    assert P.expected_affine(P.EDGE_BOXES[6][1], 4, H, W)[0, 6].max() == 0.0
    assert P.expected_affine(P.EDGE_BOXES[10][1], 4, H, W).max() == 0.0
    rois = np.array([[0, *P.SQUARE_BOX]], dtype=np.float32)
    for got in (orc.roi_align(torch.from_numpy(P.SQUARE_MAP), torch.from_numpy(rois), 0.25).numpy()[0, 0], orc.roi_align_scalar(P.SQUARE_MAP, rois, 0.25)[0, 0]):
        for (ph, pw), v in P.SQUARE_HAND.items():
            assert abs(float(got[ph, pw]) - v) < 1e-5, ((ph, pw), float(got[ph, pw]), v)
    boxes = torch.tensor([b for b, _ in P.LEVEL_EDGE_BOXES])
    assert orc.map_roi_levels(boxes).tolist() == [l for _, l in P.LEVEL_EDGE_BOXES]


def test_yaw_difference_is_taken_modulo_two_pi():
    """yaw = atan2(x, -z) has its branch cut where the gaze points straight back: two vectors 1e-5 rad apart on either side of it differ by
    2 pi in a raw subtraction (tools/parity_fuzz.py seed 11 case 891).  orc.yaw_pitch_diff wraps the yaw difference into (-pi, pi]."""
    import math
    a = torch.tensor([[1e-5, 0.3, 1.0], [0.2, 0.1, -1.0]])
    b = torch.tensor([[-1e-5, 0.3, 1.0], [0.2, 0.1, -1.0]])
    a, b = a / a.norm(dim=1, keepdim=True), b / b.norm(dim=1, keepdim=True)
    raw = (orc.yaw_pitch(a) - orc.yaw_pitch(b)).abs()
    assert abs(float(raw[0, 0]) - 2 * math.pi) < 1e-4                      # the artefact
    d = orc.yaw_pitch_diff(a, b)
    assert float(d.max()) < 3e-5 and float(d[1].max()) == 0.0
