"""-m gpu: the whole path (mcg_clip_forward through the C-ABI) against the golden vectors the
reference's own Python produced, against the oracle, and -- at BASELINE.json's full batch size --
through size-independent properties.

north_star tolerance: (yaw, pitch) of the gaze vectors within 1e-3 of the reference CPU path.
That bar is met (with margin) by the MCG_F32 engine.  The MCG_BF16 engine is the throughput
configuration BASELINE.json quotes clips/s on; its deviation is bounded by BF16_TOL below and
the measured value is printed (DESIGN.md reports it).
"""
import os

import numpy as np
import pytest
import torch

from mcgaze_amd import synth
from oracle import mcgaze_oracle as orc

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3    # north_star
BF16_TOL = 3e-1  # rad, loose bound on RANDOM-weight nets (no trained checkpoint here); the measured value is printed
CASES = ['clip224', 'clip_nonsquare', 'batch2', 'clip_t5']
KEYS = ('gaze_score', 'face_gaze_score', 'eyes_gaze_score', 'head_gaze_score')


def load_case(golden_dir, name):
    g = dict(np.load(os.path.join(golden_dir, name + '.npz')))
    B, T = int(g['B']), int(g['T'])
    ishape, pshape = tuple(int(v) for v in g['img_shape']), tuple(int(v) for v in g['pad_shape'])
    img = synth.make_clips(int(g['img_seed']), B, T, pshape[0], pshape[1])
    if ishape != pshape:
        img[:, :, ishape[0]:, :] = 0
        img[:, :, :, ishape[1]:] = 0
    return g, img, B, T, ishape


@pytest.fixture(scope='module')
def engines():
    from mcgaze_amd.engine import HipEngine
    sd = synth.make_state_dict(0)
    return {p: HipEngine(sd, precision=p) for p in ('fp32', 'bf16')}


@pytest.mark.parametrize('name', CASES)
def test_fp32_engine_matches_reference_golden(golden_dir, engines, name):
    g, img, B, T, ishape = load_case(golden_dir, name)
    N = B * T
    hw = np.tile(np.array(ishape[:2], dtype=np.int32), (N, 1))
    out = engines['fp32'].forward(torch.from_numpy(img).to('cuda:0'), T, img_hw=hw)
    torch.cuda.synchronize()
    gaze = out['gaze'].cpu()
    for i, k in enumerate(KEYS):
        d = (orc.yaw_pitch(gaze[i]) - orc.yaw_pitch(g[k])).abs().max().item()
        print(f'{name} fp32 {k}: max |d(yaw,pitch)| = {d:.2e}')
        assert d < F32_TOL, (k, d)
    boxes = out['boxes'].cpu()
    if bool(g['rescale']):
        boxes = boxes / torch.from_numpy(g['scale_factor'])[None, None, :]
    np.testing.assert_allclose(boxes.numpy(), g['det_bboxes'][..., :4], atol=5e-2, rtol=1e-4)
    np.testing.assert_allclose(out['scores'].cpu().numpy(), g['det_bboxes'][..., 4], atol=1e-3)


@pytest.mark.parametrize('name', CASES)
def test_bf16_engine_close_to_reference_golden(golden_dir, engines, name):
    g, img, B, T, ishape = load_case(golden_dir, name)
    N = B * T
    hw = np.tile(np.array(ishape[:2], dtype=np.int32), (N, 1))
    out = engines['bf16'].forward(torch.from_numpy(img).to('cuda:0'), T, img_hw=hw)
    torch.cuda.synchronize()
    gaze = out['gaze'].cpu()
    assert torch.isfinite(gaze).all()
    d = (orc.yaw_pitch(gaze[0]) - orc.yaw_pitch(g['gaze_score'])).abs()
    ang = torch.rad2deg(torch.acos((gaze[0] * torch.from_numpy(g['gaze_score'])).sum(-1).clamp(-1, 1)))
    print(f'{name} bf16 fused gaze: max |d(yaw,pitch)| = {d.max().item():.2e} rad, mean angular error = {ang.mean().item():.3f} deg')
    assert d.max().item() < BF16_TOL


def test_batched_equals_per_clip_bitwise(engines):
    """clips are independent -- a batch of B clips must reproduce B single-clip
    calls exactly (same kernels, same reduction order per output element)."""
    T, B = 7, 5
    img = torch.from_numpy(synth.make_clips(42, B, T)).to('cuda:0')
    for p in ('fp32', 'bf16'):
        e = engines[p]
        whole = {k: v.clone() for k, v in e.forward(img, T).items()}
        for b in range(B):
            part = e.forward(img[b * T:(b + 1) * T].contiguous(), T)
            torch.cuda.synchronize()
            assert torch.equal(part['gaze'], whole['gaze'][:, b * T:(b + 1) * T]), (p, b)
            assert torch.equal(part['boxes'], whole['boxes'][b * T:(b + 1) * T]), (p, b)


def test_chunked_trunk_is_bitwise_identical(engines):
    T, B = 7, 4
    img = torch.from_numpy(synth.make_clips(43, B, T)).to('cuda:0')
    e = engines['bf16']
    a = {k: v.clone() for k, v in e.forward(img, T, chunk_frames=0).items()}
    b = e.forward(img, T, chunk_frames=7)
    torch.cuda.synchronize()
    assert torch.equal(a['gaze'], b['gaze']) and torch.equal(a['boxes'], b['boxes'])


def test_full_batch_properties(engines):
    """BASELINE.json configs[2] size: 64 clips x 7 frames, bf16.  Size-independent properties:
    unit-norm outputs, finite values, clip-permutation equivariance, clip 0 equals the oracle-checked
    single-clip result."""
    T, B = 7, 64
    img = torch.from_numpy(synth.make_clips(3, B, T)).to('cuda:0')
    e = engines['bf16']
    out = {k: v.clone() for k, v in e.forward(img, T, chunk_frames=56).items()}
    torch.cuda.synchronize()
    assert torch.isfinite(out['gaze']).all() and torch.isfinite(out['boxes']).all()
    assert torch.allclose(out['gaze'].norm(dim=-1), torch.ones(4, B * T, device='cuda:0'), atol=1e-5)
    assert ((out['scores'] > 0) & (out['scores'] < 1)).all()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
    idx = (perm[:, None] * T + torch.arange(T)[None]).reshape(-1).to('cuda:0')
    out_p = e.forward(img[idx].contiguous(), T, chunk_frames=56)
    torch.cuda.synchronize()
    assert torch.equal(out_p['gaze'], out['gaze'][:, idx])
    one = e.forward(img[:T].contiguous(), T)
    assert torch.equal(one['gaze'], out['gaze'][:, :T])


def test_errors_are_loud(engines):
    from mcgaze_amd.lib import McgError
    e = engines['fp32']
    with pytest.raises(McgError):
        e.forward(torch.zeros(6, 3, 224, 224, device='cuda:0'), 7)      # N not a multiple of clip_length
    with pytest.raises(McgError):
        e.forward(torch.zeros(7, 3, 100, 224, device='cuda:0'), 7)      # H not a multiple of 32
