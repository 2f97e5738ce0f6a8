"""-m gpu: the whole path (mcg_clip_forward through the C-ABI) against the golden vectors the
reference's own Python produced, against the oracle, and -- at BASELINE.json's full batch size --
through size-independent properties.

north_star tolerance: (yaw, pitch) of the gaze vectors within 1e-3 of the reference CPU path.
That bar is met (with margin) by the MCG_F32 and MCG_F16X3 engines -- the PARITY tests.  The MCG_BF16
engine is the 16-bit throughput mode and is NOT within it on random-weight nets; the tests that
touch it are property / bound tests (finite, unit norm, bit-identical across schedules, deviation
below BF16_TOL = the largest ever measured x 1.3) and are named as such.
"""
import os

import numpy as np
import pytest
import torch

from mcgaze_amd import synth
from oracle import mcgaze_oracle as orc

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3    # north_star
# bf16 THROUGHPUT engine, RANDOM-weight nets: bounds = the largest measured deviation x 1.3 (it is NOT within north_star's 1e-3; the
# f16x3 engine is).  Goldens (224x224 clips): 0.164 rad measured; unusual shapes (64x64 frames, 101-frame clip): 0.244 rad measured.
BF16_TOL = 0.215
BF16_TOL_UNUSUAL = 0.32
# fp16 THROUGHPUT engine (MCG_F16, round 6): 11 significant bits instead of 8 -- bounds = measured x 1.3 (NOT within 1e-3 on random-weight nets either)
F16_TOL = 0.05
F16_TOL_TRAINED = 0.02
CASES = ['clip224', 'clip_nonsquare', 'batch2', 'clip_t5']
TRAINED_CASES = ['trained_clip224', 'trained_nonsquare_b2']   # synth's 'trained' weight family, goldens from the imported reference
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
    return {p: HipEngine(sd, precision=p) for p in ('fp32', 'bf16', 'f16x3', 'f16')}


@pytest.fixture(scope='module')
def engines_by_weights(engines):
    """weight seed -> (state dict, {'fp32', 'f16x3'} engines); seed 0 reuses the module's engines."""
    from mcgaze_amd.engine import HipEngine
    cache = {0: (synth.make_state_dict(0), engines)}

    def get(wseed):
        if wseed not in cache:
            sd = synth.make_state_dict(wseed)
            cache[wseed] = (sd, {p: HipEngine(sd, precision=p) for p in ('fp32', 'f16x3')})
        return cache[wseed]
    return get


PARITY_ENGINES = ['fp32', 'f16x3']   # both must meet north_star's 1e-3; f16x3 is the one bench.py times as `parity_engine`


@pytest.fixture(scope='module')
def engines_trained():
    from mcgaze_amd.engine import HipEngine
    sd = synth.make_state_dict(0, family='trained')
    return {p: HipEngine(sd, precision=p) for p in ('fp32', 'bf16', 'f16x3', 'f16')}


@pytest.mark.parametrize('precision', PARITY_ENGINES)
@pytest.mark.parametrize('name', CASES + TRAINED_CASES)
def test_fp32_engine_matches_reference_golden(golden_dir, engines, engines_trained, name, precision):
    g, img, B, T, ishape = load_case(golden_dir, name)
    if str(g.get('weight_family', 'uniform')) == 'trained':
        engines = engines_trained
    N = B * T
    hw = np.tile(np.array(ishape[:2], dtype=np.int32), (N, 1))
    out = engines[precision].forward(torch.from_numpy(img).to('cuda:0'), T, img_hw=hw)
    torch.cuda.synchronize()
    gaze = out['gaze'].cpu()
    for i, k in enumerate(KEYS):
        d = orc.yaw_pitch_diff(gaze[i], g[k]).max().item()
        print(f'{name} {precision} {k}: max |d(yaw,pitch)| = {d:.2e}')
        assert d < F32_TOL, (k, d)
    boxes = out['boxes'].cpu()
    if bool(g['rescale']):
        boxes = boxes / torch.from_numpy(g['scale_factor'])[None, None, :]
    dbox = float(np.abs(boxes.numpy() - g['det_bboxes'][..., :4]).max())
    print(f'{name} {precision} boxes: max |d| = {dbox:.2e} px (coordinates up to {float(np.abs(g["det_bboxes"][..., :4]).max()):.0f} px)')
    # boxes against the REFERENCE's: measured <= 2.0e-3 px (f16x3, the non-square golden; fp32 <= 1.1e-3) on coordinates up to 450 px, all goldens; 5e-3 px absolute
    np.testing.assert_allclose(boxes.numpy(), g['det_bboxes'][..., :4], atol=5e-3, rtol=0)
    np.testing.assert_allclose(out['scores'].cpu().numpy(), g['det_bboxes'][..., 4], atol=1e-3)


@pytest.mark.parametrize('name', CASES)
def test_bf16_engine_deviation_is_bounded_not_parity(golden_dir, engines, name):
    """A BOUND test, not a parity test: the bf16 THROUGHPUT engine is outside north_star's 1e-3 on random-weight nets;
    what is asserted is that its output is finite, unit-norm and within the largest deviation ever measured x 1.3, so that a
    regression of the 16-bit path (a wrong rounding, a swapped operand) still fails.  Parity is asserted for fp32 and f16x3 above."""
    g, img, B, T, ishape = load_case(golden_dir, name)
    N = B * T
    hw = np.tile(np.array(ishape[:2], dtype=np.int32), (N, 1))
    out = engines['bf16'].forward(torch.from_numpy(img).to('cuda:0'), T, img_hw=hw)
    torch.cuda.synchronize()
    gaze = out['gaze'].cpu()
    assert torch.isfinite(gaze).all()
    d = orc.yaw_pitch_diff(gaze[0], g['gaze_score'])
    ang = torch.rad2deg(torch.acos((gaze[0] * torch.from_numpy(g['gaze_score'])).sum(-1).clamp(-1, 1)))
    print(f'{name} bf16 fused gaze: max |d(yaw,pitch)| = {d.max().item():.2e} rad, mean angular error = {ang.mean().item():.3f} deg')
    assert d.max().item() < BF16_TOL


@pytest.mark.parametrize('name', TRAINED_CASES)
def test_bf16_engine_on_the_trained_weight_family_is_reported(golden_dir, engines_trained, name):
    """VERDICT r3 item 5c asked whether a 16-bit arithmetic could be parity-grade on REALISTIC boxes (inside the frame, small regression
    heads): measured here on the 'trained' weight family and printed; bounded loosely (a regression guard), never a parity claim."""
    g, img, B, T, ishape = load_case(golden_dir, name)
    hw = np.tile(np.array(ishape[:2], dtype=np.int32), (B * T, 1))
    out = engines_trained['bf16'].forward(torch.from_numpy(img).to('cuda:0'), T, img_hw=hw)
    torch.cuda.synchronize()
    d = orc.yaw_pitch_diff(out['gaze'].cpu()[0], g['gaze_score']).max().item()
    print(f'{name} bf16 on the trained family: max |d(yaw,pitch)| = {d:.2e} rad (north_star: 1e-3)')
    assert np.isfinite(d) and d < BF16_TOL_UNUSUAL


@pytest.mark.parametrize('name', CASES + TRAINED_CASES)
def test_f16_engine_deviation_is_bounded_and_below_bf16(golden_dir, engines, engines_trained, name):
    """A BOUND test, not a parity test (like the bf16 one above): MCG_F16 stores activations and weights as fp16 -- bf16's kernels with 11
    significant bits instead of 8.  Asserted: finite, unit-norm, inside the measured bound x 1.3; printed beside the bf16 engine's deviation
    on the same golden (the model's discontinuities make single inputs noisy: 'closer than bf16' is asserted on the SUM over the goldens in
    test_f16_is_closer_than_bf16_over_the_goldens)."""
    g, img, B, T, ishape = load_case(golden_dir, name)
    trained = str(g.get('weight_family', 'uniform')) == 'trained'
    eng = engines_trained if trained else engines
    hw = np.tile(np.array(ishape[:2], dtype=np.int32), (B * T, 1))
    d = {}
    for p in ('f16', 'bf16'):
        out = eng[p].forward(torch.from_numpy(img).to('cuda:0'), T, img_hw=hw)
        torch.cuda.synchronize()
        gaze = out['gaze'].cpu()
        assert torch.isfinite(gaze).all(), p
        assert float((gaze.norm(dim=-1) - 1).abs().max()) < 1e-5, p
        d[p] = orc.yaw_pitch_diff(gaze[0], g['gaze_score']).max().item()
    print(f'{name}: max |d(yaw,pitch)| vs the reference golden: f16 {d["f16"]:.2e} rad, bf16 {d["bf16"]:.2e} rad (north_star: 1e-3)')
    assert d['f16'] < (F16_TOL_TRAINED if trained else F16_TOL)


def test_f16_is_closer_than_bf16_over_the_goldens(golden_dir, engines, engines_trained):
    tot = {'f16': 0.0, 'bf16': 0.0}
    for name in CASES + TRAINED_CASES:
        g, img, B, T, ishape = load_case(golden_dir, name)
        eng = engines_trained if str(g.get('weight_family', 'uniform')) == 'trained' else engines
        hw = np.tile(np.array(ishape[:2], dtype=np.int32), (B * T, 1))
        for p in tot:
            out = eng[p].forward(torch.from_numpy(img).to('cuda:0'), T, img_hw=hw)
            torch.cuda.synchronize()
            tot[p] += float(orc.yaw_pitch_diff(out['gaze'].cpu()[0], g['gaze_score']).mean())
    print(f'sum over the goldens of mean |d(yaw,pitch)|: f16 {tot["f16"]:.3e}, bf16 {tot["bf16"]:.3e} rad')
    assert tot['f16'] < 0.5 * tot['bf16']


def test_f16_kernel_variants_are_bit_identical(engines):
    """The specialised 16-bit TRUNK kernels (conv3x3_c64, the fused stem, pw_pair, pw_single) are templates over the number format since
    round 6; their fp16 instantiations must keep the property the bf16 ones are tested for: every variant gives the bits of the generic
    contraction kernel.  Options toggled one at a time on the f16 engine, pyramid and outputs compared."""
    e = engines['f16']
    T = 7
    for shape in ((85, 224, 224), (14, 96, 160)):
        img = torch.from_numpy(synth.make_clips(73, 1, *shape)).to('cuda:0')
        n = (shape[0] // T) * T
        ref_p = [p.clone() for p in e.backbone_fpn(img)]
        ref_o = {k: v.clone() for k, v in e.forward(img[:n], T).items()}
        for opt in ('conv3x3_c64', 'stem_fused', 'pointwise_pair', 'pointwise_stream'):   # (decoder_chain: tests/test_gpu_kernels.py::test_mlp_chain_matches_unfused_bitwise)
            e.set_option(opt, 0)
            try:
                out_p = e.backbone_fpn(img)
                out_o = e.forward(img[:n], T)
                torch.cuda.synchronize()
                for lvl, (a, b) in enumerate(zip(ref_p, out_p)):
                    assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (shape, opt, lvl)
                for k in ref_o:
                    assert torch.equal(ref_o[k], out_o[k]), (shape, opt, k)
            finally:
                e.set_option(opt, 1)


def test_batched_equals_per_clip_bitwise(engines):
    """clips are independent -- a batch of B clips must reproduce B single-clip
    calls exactly (same kernels, same reduction order per output element)."""
    T, B = 7, 5
    img = torch.from_numpy(synth.make_clips(42, B, T)).to('cuda:0')
    for p in ('fp32', 'bf16', 'f16x3', 'f16'):
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


@pytest.mark.parametrize('precision', ['f16x3', 'bf16', 'f16'])
def test_full_batch_properties(engines, precision):
    """BASELINE.json configs[2] size: 64 clips x 7 frames, the headline engine (f16x3: fused bottleneck tails, streaming 1x1 kernels,
    row-block chains -- every kernel at the size bench.py times) and the bf16 engine.  Size-independent properties: unit-norm
    outputs, finite values, clip-permutation equivariance, clip 0 equals the oracle-checked single-clip result bit for bit."""
    T, B = 7, 64
    img = torch.from_numpy(synth.make_clips(3, B, T)).to('cuda:0')
    e = engines[precision]
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


def test_registry_surface_reproduces_reference_outputs(golden_dir):
    """The drop-in boundary end to end: config -> init_detector -> model(return_loss=False, ...) returns the
    reference's output structure and values (fp32 engine) for single clips, and for a batch via clip_length."""
    from mcgaze_amd import init_detector
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = init_detector(os.path.join(root, 'configs', 'mcgaze', 'r50_clip7_gaze360.py'), None, device='cuda:0', precision='fp32')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(0).items()}, strict=True)
    assert model.CLASSES == ('face', 'eyes', 'head') and model.cfg.clip_length == 7
    for name in ('clip224', 'clip_nonsquare', 'batch2'):
        g, img, B, T, ishape = load_case(golden_dir, name)
        pshape = tuple(int(v) for v in g['pad_shape'])
        metas = synth.make_img_metas(B * T, ishape, pshape, tuple(float(v) for v in g['scale_factor']))
        kw = dict(clip_length=T) if B > 1 else {}
        with torch.no_grad():
            (det_bboxes, det_labels), gaze = model(img=[torch.from_numpy(img)], img_metas=[metas], return_loss=False,
                                                   rescale=bool(g['rescale']), format=False, **kw)
        assert len(det_bboxes) == B * T and all(tuple(d.shape) == (3, 5) for d in det_bboxes) and det_labels[0] == [0, 1, 2]
        assert set(gaze) == set(KEYS) and all(tuple(v.shape) == (B * T, 3) and v.is_cuda for v in gaze.values())
        for k in KEYS:
            assert orc.yaw_pitch_diff(gaze[k].cpu(), g[k]).max().item() < F32_TOL
        np.testing.assert_allclose(torch.stack(det_bboxes).cpu().numpy(), g['det_bboxes'], atol=5e-2, rtol=1e-4)
    # format=True goes through bbox2result: per-frame list of per-class arrays
    g, img, B, T, ishape = load_case(golden_dir, 'clip224')
    res, _ = model(img=[torch.from_numpy(img)], img_metas=[synth.make_img_metas(T)], return_loss=False, rescale=False, format=True)
    assert len(res) == T and len(res[0]) == 3 and res[0][0].shape == (1, 5)


def test_pipelined_runner_matches_serial_engine(engines):
    """The two-stream batch pipeline must produce exactly what the serial single-stream path produces, batch by batch."""
    from mcgaze_amd.engine import PipelinedRunner
    T, B = 7, 6
    e = engines['bf16']
    batches = [torch.from_numpy(synth.make_clips(100 + i, B, T)).to('cuda:0') for i in range(5)]
    serial = [{k: v.clone() for k, v in e.forward(x, T).items()} for x in batches]
    runner = PipelinedRunner(e, B * T, 224, 224, T)
    outs = [dict(gaze=torch.zeros(4, B * T, 3, device='cuda:0'), boxes=torch.zeros(B * T, 3, 4, device='cuda:0'),
                 scores=torch.zeros(B * T, 3, device='cuda:0')) for _ in batches]
    for x, o in zip(batches, outs):
        runner.submit(x, o)
    runner.flush()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(serial, outs)):
        for k in ('gaze', 'boxes', 'scores'):
            assert torch.equal(a[k], b[k]), (i, k)


@pytest.mark.parametrize('precision', ['f16x3', 'bf16'])
def test_pipelined_schedule_soak(engines, precision):
    """The production schedule (bench.py: two trunk frame ranges on concurrent streams, the decoder of batch k beside the trunk of batch
    k + 1) run long enough to catch a rare timing-dependent fault: 120 batches cycling over three inputs, in groups of four between
    drains, EVERY output compared bit for bit with the serial single-stream result of its input.  (The fused tail's ring race --
    profiles/r03_x_lds_war.md -- showed once in ~200 launches under another engine's load; this is the same kind of net under the
    engine's own concurrency.)"""
    from mcgaze_amd.engine import PipelinedRunner
    T, B = 7, 16
    e = engines[precision]
    batches = [torch.from_numpy(synth.make_clips(300 + i, B, T)).to('cuda:0') for i in range(3)]
    e.set_option('trunk_streams', 1)
    serial = [{k: v.clone() for k, v in e.forward(x, T).items()} for x in batches]
    e.set_option('trunk_streams', 2)
    torch.cuda.synchronize()
    runner = PipelinedRunner(e, B * T, 224, 224, T)
    ring = [dict(gaze=torch.zeros(4, B * T, 3, device='cuda:0'), boxes=torch.zeros(B * T, 3, 4, device='cuda:0'),
                 scores=torch.zeros(B * T, 3, device='cuda:0')) for _ in range(4)]
    bad = []
    for group in range(30):
        for j in range(4):
            runner.submit(batches[(4 * group + j) % 3], ring[j])
        runner.flush()
        torch.cuda.synchronize()
        for j in range(4):
            for k in ('gaze', 'boxes', 'scores'):
                if not torch.equal(ring[j][k], serial[(4 * group + j) % 3][k]):
                    bad.append((4 * group + j, k))
    assert not bad, bad[:8]


def test_bench_host_input_leg_delivers_the_forward_results(engines):
    """bench.py `host_input` (the PCIe-inclusive rate): batches from pinned host memory through the upload stream, the batch pipeline and
    the return stream -- what lands in host memory must be, bit for bit, a plain forward of what was uploaded (both input formats)."""
    import bench
    out = bench.host_input_leg(engines['f16x3'], torch.device('cuda', 0), 6, 7, 224, steps=5, warmup=3)
    for mode in ('f32', 'uint8'):
        assert out[mode]['verified'] is True and out[mode]['value'] > 0, (mode, out[mode])


@pytest.mark.parametrize('B,T,H,W', [(1, 33, 64, 96), (3, 1, 96, 64), (1, 2, 448, 448), (1, 101, 64, 64)])
def test_fp32_engine_matches_oracle_on_unusual_shapes(engines, B, T, H, W):
    """Shapes the reference supports but the goldens do not cover: long clips (33 frames; 101, the longest the demo feeds:
    SURVEY.md section 8(b)), single-frame clips, and the L2CS input size 448x448 -- fp32 engine vs the oracle at north_star's
    tolerance (the oracle itself is pinned to the reference by tests/test_oracle.py)."""
    sd = synth.make_state_dict(0)
    img = synth.make_clips(100 + T, B, T, H, W)
    metas = synth.make_img_metas(B * T, (H, W, 3))
    want_det, want_gaze = orc.forward(sd, img, metas, T)
    for precision in PARITY_ENGINES:
        out = engines[precision].forward(torch.from_numpy(img).to('cuda:0'), T)
        torch.cuda.synchronize()
        for i, k in enumerate(KEYS):
            d = orc.yaw_pitch_diff(out['gaze'][i].cpu(), want_gaze[k]).max().item()
            assert d < F32_TOL, (precision, k, d)
        np.testing.assert_allclose(out['boxes'].cpu().numpy(), want_det[..., :4].numpy(), atol=5e-2, rtol=1e-4)
    bf = engines['bf16'].forward(torch.from_numpy(img).to('cuda:0'), T)
    assert torch.isfinite(bf['gaze']).all() and orc.yaw_pitch_diff(bf['gaze'][0].cpu(), want_gaze['gaze_score']).max().item() < BF16_TOL_UNUSUAL


def test_frame_range_cap_and_stream_split_do_not_change_results(engines):
    """The trunk runs as concurrent frame ranges, capped so that no activation outgrows the 2 GiB descriptor window; a batch
    beyond k capped ranges takes several rounds.  Forcing a tiny cap (many rounds), one stream, or four must all give the bits
    of the default (options through mcg_engine_set_option)."""
    T, B = 7, 40
    img = torch.from_numpy(synth.make_clips(77, B, T, 64, 96)).to('cuda:0')
    e = engines['bf16']
    ref = {k: v.clone() for k, v in e.forward(img, T).items()}
    try:
        for opts in (dict(max_range_frames=40), dict(trunk_streams=1), dict(trunk_streams=4), dict(trunk_streams=3, max_range_frames=57)):
            e.set_option('trunk_streams', 2); e.set_option('max_range_frames', 0)
            for k, v in opts.items():
                e.set_option(k, v)
            out = e.forward(img, T)
            torch.cuda.synchronize()
            for k in ref:
                assert torch.equal(out[k], ref[k]), (opts, k)
    finally:
        e.set_option('trunk_streams', 2); e.set_option('max_range_frames', 0)


# Per-stage deviation of the bf16 THROUGHPUT engine from the reference goldens (random weights), as a fraction of each tensor's
# scale.  What this test established: with IDENTICAL boxes going in (stage 0) bf16 rounding keeps the query features within ~1 % of
# scale and the refined boxes within ~1 px.  The engine's 0.03-0.24 rad gaze deviation comes from DISCONTINUITIES of the model
# itself, which a one-pixel box perturbation triggers in the later stages: (1) a box whose sqrt(area) sits next to a pyramid-level
# boundary (56 * 2^k px, single_level_roi_extractor.py:51-54) is routed to another FPN level; (2) RoIAlign drops a sample to zero
# when it leaves [-1, H] x [-1, W] (mmcv roi_align: `y < -1 || y > height -> 0`) and the random-weight boxes grow far beyond the
# frame.  After either, a stage's features differ by 20-50 % of scale.  So the smooth bound (measured x 1.3) is asserted where the
# inputs are identical, the later stages are reported with the routing flips that explain them, and the parity-grade engine must
# follow the reference through every stage, routing included.
BF16_SMOOTH_OBJ_BOUND = 0.02     # measured 0.010 on stage 0 of clip224
BF16_SMOOTH_BOX_BOUND = 2.0      # px; measured 1.0


@pytest.mark.parametrize('name', ['clip224', 'batch2'])
def test_bf16_error_growth_per_stage(golden_dir, engines, name):
    from mcgaze_amd import engine as E
    g, img, B, T, ishape = load_case(golden_dir, name)
    e = engines['bf16']
    pyr = e.backbone_fpn(torch.from_numpy(img).to('cuda:0'))
    for i, p in enumerate(pyr):      # pyramid samples the reference recorded
        flat = p.permute(0, 3, 1, 2).reshape(-1).float().cpu()
        err = float((flat[torch.from_numpy(g[f'fpn{i}_idx'])] - torch.from_numpy(g[f'fpn{i}_val'])).abs().max() / float(g[f'fpn{i}_absmean']))
        print(f'{name} bf16 P{i + 2}: max sample error = {err:.3f} of the level\'s mean |value|')
        assert err < 0.045, (i, err)          # measured 0.024 .. 0.032
    N = B * T
    boxes = torch.from_numpy(g['init_boxes']).to('cuda:0')
    ref_boxes_in = torch.from_numpy(g['init_boxes'])
    obj = e.weights.init_feats[None].expand(N, 3, 256).contiguous()
    flipped = False
    for s in range(4):
        roi, lv = E.roi_align(pyr, boxes)
        flips = int((lv.cpu().long() != orc.map_roi_levels(ref_boxes_in.reshape(-1, 4))).sum())
        flipped = flipped or flips > 0
        obj, boxes, cls = E.stage_forward(e.weights.stages[s], roi, obj, boxes, T)
        torch.cuda.synchronize()
        want = torch.from_numpy(g['stage_obj'][s])
        err = float((obj.float().cpu() - want).abs().max() / want.abs().max())
        berr = float((boxes.cpu() - torch.from_numpy(g['stage_boxes'][s])).abs().max())
        print(f'{name} bf16 stage {s}: obj error = {err:.4f} of scale, boxes max |d| = {berr:.2f} px, level-routing flips so far: {flips}{" (discontinuity)" if flipped else ""}')
        assert np.isfinite(err) and np.isfinite(berr)
        if s == 0:
            assert flips == 0 and err < BF16_SMOOTH_OBJ_BOUND and berr < BF16_SMOOTH_BOX_BOUND, (s, err, berr)
        ref_boxes_in = torch.from_numpy(g['stage_boxes'][s])
    # the parity-grade engine follows the reference through every stage, routing included
    e3 = engines['f16x3']
    pyr3 = e3.backbone_fpn(torch.from_numpy(img).to('cuda:0'))
    boxes3, obj3 = torch.from_numpy(g['init_boxes']).to('cuda:0'), e3.weights.init_feats[None].expand(N, 3, 256).contiguous()
    ref_boxes_in = torch.from_numpy(g['init_boxes'])
    for s in range(4):
        roi, lv = E.roi_align(pyr3, boxes3)
        assert int((lv.cpu().long() != orc.map_roi_levels(ref_boxes_in.reshape(-1, 4))).sum()) == 0, s
        obj3, boxes3, _ = E.stage_forward(e3.weights.stages[s], roi, obj3, boxes3, T, split=True)
        torch.cuda.synchronize()
        want = torch.from_numpy(g['stage_obj'][s])
        err = float((obj3.cpu() - want).abs().max() / want.abs().max())
        print(f'{name} f16x3 stage {s}: obj error = {err:.2e} of scale')
        assert err < 2e-5, (s, err)       # bf16 halves (the first version): < 2e-4
        ref_boxes_in = torch.from_numpy(g['stage_boxes'][s])


@pytest.mark.parametrize('precision', ['bf16', 'f16x3'])
def test_bench_schedule_is_bitwise_equal_to_per_clip_calls(precision):
    """Exactly what bench.py times at BASELINE.json configs[2]: 64 clips x 7 x 3 x 224 x 224 (448 frames), chunk_frames = 0 ->
    two concurrent frame ranges on probed side streams, the two-deep PipelinedRunner, and the result exchange on its own stream
    through a (single-rank) RCCL group -- driven through bench.Leg itself.  Every clip of the timed schedule's output must
    equal, bit for bit, a separate single-clip call of the same engine, and bench's own `verify()` must agree."""
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv, sys.argv = sys.argv, ['bench.py']
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    dev = torch.device('cuda', 0)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29561')
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        B, T = 64, 7
        img = torch.from_numpy(synth.make_clips(3, B, T)).to(dev)
        leg = bench.Leg(a, precision, dev, 1, 0, dist, img, B, T)
        assert leg.runner is not None and leg.comm is not None
        for _ in range(5):
            leg.step()
        leg.drain()
        torch.cuda.synchronize()
        assert leg.verify() is True
        for slot in (0, 1):
            out = leg.outs[slot]
            merged = leg.gathers[slot].all_single()      # what the exchange delivered
            assert torch.equal(merged, leg.gathers[slot].local)
            for b in range(0, B, 3):
                one = leg.eng.forward(img[b * T:(b + 1) * T].contiguous(), T)
                torch.cuda.synchronize()
                assert torch.equal(one['gaze'], out['gaze'][:, b * T:(b + 1) * T]), (slot, b)
                assert torch.equal(one['boxes'], out['boxes'][b * T:(b + 1) * T]), (slot, b)
                assert torch.equal(one['scores'], out['scores'][b * T:(b + 1) * T]), (slot, b)
    finally:
        if own_group:
            dist.destroy_process_group()


@pytest.mark.parametrize('precision', ['bf16', 'f16x3'])
def test_graphed_forward_is_bit_identical(engines, precision):
    """The HIP-graph latency path (engine.GraphedForward: one graph launch per clip) against the eager call, on several clips and
    with per-frame img_shape; also a 16-clip batch (two concurrent frame ranges inside the graph: fork / join events are captured)."""
    from mcgaze_amd.engine import GraphedForward
    e = engines[precision]
    T = 7
    g = GraphedForward(e, T, 224, 224, T, with_img_hw=True)
    for seed in (1, 2, 3):
        img = torch.from_numpy(synth.make_clips(900 + seed, 1, T)).to('cuda:0')
        hw = np.tile(np.array([200 + seed, 210], dtype=np.int32), (T, 1))
        want = {k: v.clone() for k, v in e.forward(img, T, img_hw=hw).items()}
        got = g(img, img_hw=hw)
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(got[k], want[k]), (seed, k)
    B = 16
    gb = GraphedForward(e, B * T, 224, 224, T)
    img = torch.from_numpy(synth.make_clips(77, B, T)).to('cuda:0')
    want = {k: v.clone() for k, v in e.forward(img, T).items()}
    got = gb(img)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_pointwise_pair_fusion_is_bit_identical(engines):
    """pw_pair.hpp (layer1 / layer2: a block's conv3 (+ downsample / + residual) and the next block's conv1 as one kernel) against
    the layer-granular launches: the pyramid -- hence everything downstream -- must not change by a bit, on frame sizes whose
    pixel counts are and are not multiples of the kernel's 64-pixel tile."""
    e = engines['bf16']
    try:
        for shape in ((3, 224, 224), (2, 96, 160), (5, 32, 32), (1, 448, 448)):
            img = torch.from_numpy(synth.make_clips(61, 1, *shape)).to('cuda:0')
            e.set_option('pointwise_pair', 0)
            ref = [p.clone() for p in e.backbone_fpn(img)]
            e.set_option('pointwise_pair', 1)
            out = e.backbone_fpn(img)
            torch.cuda.synchronize()
            for lvl, (a, b) in enumerate(zip(ref, out)):
                assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (shape, lvl)
    finally:
        e.set_option('pointwise_pair', 1)


@pytest.mark.parametrize('precision', ['f16x3', 'bf16'])
def test_backbone_only_matches_the_oracle(engines, precision):
    """BASELINE.json configs[1] -- R-50 backbone only, 32 clips x 7 frames x 3 x 224 x 224 -- through mcg_bench_backbone_forward
    (`bench.py`'s `backbone` sub-object times exactly this call): C2..C5 against oracle.resnet on the same 224 frames.  f16x3 within 2e-5
    of each level's scale (the parity engine; fused bottleneck tails included), bf16 within 6 % (bound test: 8-bit mantissas over 50
    layers); and the call leaves the engine usable (a full forward afterwards reproduces itself bit for bit)."""
    e = engines[precision]
    B, T = 32, 7
    img = synth.make_clips(21, B, T)
    x = torch.from_numpy(img).to('cuda:0')
    before = {k: v.clone() for k, v in e.forward(x[:T].contiguous(), T).items()}
    torch.set_num_threads(16)
    want = orc.resnet(orc.as_torch(synth.make_state_dict(0)), torch.from_numpy(img))
    e.set_option('trunk_streams', 1)
    try:
        got = e.backbone_only(x, return_levels=True)
        torch.cuda.synchronize()
        for i, (g, w) in enumerate(zip(got, want)):
            err = float((g.float().permute(0, 3, 1, 2).cpu() - w).abs().max() / w.abs().max())
            print(f'backbone only, {precision}, C{i + 2}: max error {err:.2e} of scale')
            assert err < (2e-5 if precision == 'f16x3' else 6e-2), (precision, i, err)
    finally:
        e.set_option('trunk_streams', 2)
    after = e.forward(x[:T].contiguous(), T)
    torch.cuda.synchronize()
    assert all(torch.equal(before[k], after[k]) for k in before)


def test_fused_bottleneck_tails_match_the_layer_granular_trunk(engines):
    """The f16x3 engine with layer1 through bneck_x3.hpp (default) against the same engine with the layer-granular launches
    (`bottleneck_fused` = 0).  Not bit-identical by construction -- the chained contractions visit the 16 channels of a K-step in a
    permuted order -- but the same products in f32: every pyramid level within 2e-6 of its scale, the gaze within 1e-5 rad, and the
    fused path itself bit-identical between a batch and its clips (test_batched_equals_per_clip_bitwise covers that for the default)."""
    e = engines['f16x3']
    try:
        for shape in ((3, 224, 224), (2, 96, 160), (5, 32, 32), (1, 448, 448)):
            img = torch.from_numpy(synth.make_clips(61, 1, *shape)).to('cuda:0')
            e.set_option('bottleneck_fused', 0)
            ref = [p.clone() for p in e.backbone_fpn(img)]
            gref = e.forward(img, shape[0])['gaze'].clone()
            e.set_option('bottleneck_fused', 1)
            out = e.backbone_fpn(img)
            gout = e.forward(img, shape[0])['gaze']
            torch.cuda.synchronize()
            for lvl, (a, b) in enumerate(zip(ref, out)):
                err = float((a - b).abs().max() / a.abs().max())
                assert 0 < err < 2e-6 or (err == 0 and shape[1] < 0), (shape, lvl, err)   # err > 0: the fused kernel really ran
            ang = 2 * torch.asin(((gout.double() - gref.double()).norm(dim=-1) / 2).clamp(max=1))
            assert float(ang.max()) < 1e-5, (shape, float(ang.max()))
    finally:
        e.set_option('bottleneck_fused', 1)


def test_blocked_tensors_between_fused_tails_change_no_bit(engines):
    """`bottleneck_blocked` (default 1): a tensor that only travels from one fused bottleneck tail to the next (the residual inside layer1 /
    layer2) is stored in the kernel's blocked layout -- whole-line stores and loads.  A layout, not an arithmetic: pyramid and outputs must
    be bit-identical to the engine with [M][C] tensors throughout, also where the 8 x 28 tiles do not tile the map (ragged: 96x160 gives
    24x40 and 12x20 maps; 32x32 gives 8x8 and 4x4) and where the blocked grid would not fit its buffer (the engine then keeps [M][C])."""
    e = engines['f16x3']
    try:
        for shape in ((3, 224, 224), (2, 96, 160), (5, 32, 32), (1, 448, 448), (4, 224, 256), (9, 64, 352)):
            img = torch.from_numpy(synth.make_clips(67, 1, *shape)).to('cuda:0')
            e.set_option('bottleneck_blocked', 0)
            ref = [p.clone() for p in e.backbone_fpn(img)]
            oref = {k: v.clone() for k, v in e.forward(img, shape[0]).items()}
            e.set_option('bottleneck_blocked', 1)
            out = e.backbone_fpn(img)
            oout = e.forward(img, shape[0])
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(ref, out)), shape
            assert all(torch.equal(oref[k], oout[k]) for k in oref), shape
    finally:
        e.set_option('bottleneck_blocked', 1)


def test_winograd_tile_fallback_changes_no_bit(engines):
    """`wino_tile` (default -1: by grid size): the one-wave-per-SIMD F(2,3) tile keeps hand-issued register loads in flight (its build is
    gated on zero spills, csrc/check_resources.py -- ADVICE r5); tile 0, the 8-wave kernel with its weights through LDS, is the run-time
    fallback.  Every tile computes the same bits: pyramid and outputs through the engine must not change, on a batch large enough for the
    default to pick wino_x3w_kernel (63 frames) and on a single clip."""
    e = engines['f16x3']
    try:
        for shape in ((63, 224, 224), (7, 224, 224), (14, 96, 160)):
            img = torch.from_numpy(synth.make_clips(71, 1, *shape)).to('cuda:0')
            e.set_option('wino_tile', -1)
            ref = [p.clone() for p in e.backbone_fpn(img)]
            oref = {k: v.clone() for k, v in e.forward(img, 7).items()}
            for tile in (0, 3):
                e.set_option('wino_tile', tile)
                out = e.backbone_fpn(img)
                oout = e.forward(img, 7)
                torch.cuda.synchronize()
                assert all(torch.equal(a, b) for a, b in zip(ref, out)), (shape, tile)
                assert all(torch.equal(oref[k], oout[k]) for k in oref), (shape, tile)
    finally:
        e.set_option('wino_tile', -1)
    with pytest.raises(Exception):
        e.set_option('wino_tile', 4)


def test_pointwise_stream_kernel_is_bit_identical(engines):
    """pw_single.hpp (layer2's 1x1 convs and the P2 lateral as a persistent kernel with register-resident weights) against the
    generic contraction kernel: the pyramid must not change by a bit.  The kernel takes over from 64 Ki output pixels: 85 frames of
    224x224 give layer2 66640 pixels (not a multiple of the 32-pixel tile), 22 frames of 224x256 reach only the lateral, 340 frames
    bring in layer3 (channel-split workgroups) with 66640 pixels."""
    e = engines['bf16']
    try:
        for shape in ((85, 224, 224), (22, 224, 256), (96, 224, 224), (340, 224, 224)):
            img = torch.from_numpy(synth.make_clips(67, 1, *shape)).to('cuda:0')
            e.set_option('pointwise_stream', 0)
            ref = [p.clone() for p in e.backbone_fpn(img)]
            e.set_option('pointwise_stream', 1)
            out = e.backbone_fpn(img)
            torch.cuda.synchronize()
            for lvl, (a, b) in enumerate(zip(ref, out)):
                assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (shape, lvl)
    finally:
        e.set_option('pointwise_stream', 1)


# The three inputs of the 2 400-case fuzz (tools/parity_fuzz.py, seeds 2 / 7 / 11) on which the f16x3 engine leaves 1e-3 rad on (yaw, pitch),
# pinned by name: what IS true of them is asserted, so a build that makes them worse fails here and not only in a
# 20-minute fuzz run.  (fuzz seed, case, kind, bound on the angle between the engine's and the oracle's gaze vectors)
KNOWN_FUZZ_EXCEPTIONS = [
    (7, 502, 'pole', 4e-5),        # oracle gaze y = +0.9999: yaw = atan2(x, -z) amplifies a 3e-5 rad vector error to 1.0e-3 on the yaw
    (11, 279, 'pole', 4e-5),       # oracle gaze y = +0.9996: 1.08e-3 on the yaw from a 3.1e-5 rad vector error
    (11, 721, 'validity', 5e-3),   # one RoIAlign sample on the other side of its validity edge (box 1e-3 px from the oracle's): 4.5e-3 rad
]


@pytest.mark.gpu
@pytest.mark.parametrize('seed,case,kind,bound', KNOWN_FUZZ_EXCEPTIONS)
def test_known_fuzz_exceptions_stay_what_they_are(seed, case, kind, bound):
    """profiles/r04_r_parity_fuzz_final.md: 2 397 of 2 400 random inputs within 1e-3 rad.  The two pole inputs must keep their gaze VECTOR
    within 4e-5 rad of the oracle's with no level / sample-validity flip anywhere in the chain (the 1e-3 on the yaw is the coordinate
    singularity, not the arithmetic); the validity-edge input may cross its ONE discontinuity and must stay within 5e-3 rad.  The fp32
    engine is inside 1e-3 on all three."""
    from mcgaze_amd.engine import HipEngine
    from tests import parity_tools as PT
    k = synth.fuzz_case(seed, case)
    sd = synth.make_state_dict(k['wseed'])
    stages = []
    _, ref = orc.forward(sd, k['img'], k['metas'], k['T'], collect=stages)
    want = orc.yaw_pitch(ref['gaze_score'])
    N = k['B'] * k['T']
    hw = None if k['full'] else np.tile(np.array(k['img_shape'], dtype=np.int32), (N, 1))
    for prec in ('f16x3', 'fp32'):
        eng = HipEngine(sd, precision=prec)
        out = eng.forward(torch.from_numpy(k['img']).cuda(), k['T'], img_hw=hw)
        g = out['gaze'][0].cpu()
        ang = float((2 * torch.asin(((g.double() - ref['gaze_score'].double()).norm(dim=-1) / 2).clamp(max=1))).max())
        d = float(orc.wrap_yaw(orc.yaw_pitch(g) - want).abs().max())
        print(f'fuzz seed {seed} case {case} ({kind}) {prec}: max d(yaw, pitch) {d:.3e} rad, max angle between gaze vectors {ang:.3e} rad')
        if prec == 'fp32':
            assert d <= 1e-3, (prec, d)
            continue
        assert ang <= bound, (prec, ang)
        rep = PT.stage_report(eng, prec, sd, k['img'], k['metas'], k['T'], stages)
        flips = sum(a + v for a, v, _, _ in rep['free_running'])
        assert max(rep['teacher_forced']) < 1e-4, rep['teacher_forced']       # every stage, fed the oracle's inputs, is parity-grade
        if kind == 'pole':
            assert flips == 0 and d <= 2e-3, (flips, d)                           # no discontinuity; the yaw error is the pole's amplification
            assert float(ref['gaze_score'][:, 1].abs().max()) > 0.999             # ... and the input IS a pole input
        else:
            assert flips <= 1 and d <= 5e-3, (flips, d)


FUZZ_CASES = 16
FUZZ_MAX_CROSSINGS = {'fp32': 0, 'f16x3': 1}   # asserted RATE over the 16 cases: profiles/r02_m_parity_fuzz.md measured 0 / 2400 (fp32) and 1 / 2400 (f16x3)
FUZZ_CROSSED_CLIP_BOUND = 0.05                 # rad, gaze-vector angle inside a clip whose chain crossed a discontinuity (the one observed: 4.5e-3)
_fuzz_seen = {}


def test_pointwise_stream_x3_matches_the_contraction_kernel(engines):
    """pw_single_x3.hpp (f16x3: the P2 lateral and layer3's conv3 as a persistent kernel with register-resident split weights) against the
    generic x3 contraction kernel: same K order, term order and rounding points -- the pyramid must not change by a bit.  The kernel takes
    over from 64 Ki output pixels: 22 frames of 224 x 256 reach only the lateral (ragged tile count), 340 frames bring in layer3."""
    e = engines['f16x3']
    try:
        for shape in ((22, 224, 256), (96, 224, 224), (340, 224, 224)):
            img = torch.from_numpy(synth.make_clips(67, 1, *shape)).to('cuda:0')
            e.set_option('pointwise_stream', 0)
            ref = [p.clone() for p in e.backbone_fpn(img)]
            e.set_option('pointwise_stream', 1)
            out = e.backbone_fpn(img)
            torch.cuda.synchronize()
            for lvl, (a, b) in enumerate(zip(ref, out)):
                assert torch.equal(a, b), (shape, lvl, float((a - b).abs().max()))
    finally:
        e.set_option('pointwise_stream', 1)


@pytest.mark.parametrize('index', range(FUZZ_CASES))
def test_parity_on_random_shapes_and_weights(engines_by_weights, index):
    """tools/parity_fuzz.py's cases 0..15 of seed 2 (random clip length, batch, frame size, img_shape inside the padded frame, three
    weight seeds) for the two parity-grade engines.  The model has discontinuities (a RoIAlign sample leaving [-1, L], a box crossing
    a pyramid-level boundary) and (yaw, pitch) is singular at the poles, so what is asserted is what CAN hold for every input
    (tests/parity_tools.py):
      * every stage's arithmetic on the oracle's own inputs within 2e-5 of scale -- always;
      * end to end, for every CLIP of the batch whose chain crossed no discontinuity (clips are independent): gaze VECTORS within
        2e-4 rad and (yaw, pitch) within north_star's 1e-3 away from the poles;
      * for a clip that did cross one: the gaze vectors still within FUZZ_CROSSED_CLIP_BOUND (a crossing moves one sample or one
        level, it does not derail the clip), and the crossing is COUNTED -- test_parity_fuzz_discontinuity_rate below asserts the
        rate over the 16 cases (no case is skipped)."""
    from tests import parity_tools as PT
    k = synth.fuzz_case(2, index)
    sd, engs = engines_by_weights(k['wseed'])
    stages = []
    _, ref = orc.forward(sd, k['img'], k['metas'], k['T'], collect=stages)
    N = k['B'] * k['T']
    hw = None if k['full'] else np.tile(np.array(k['img_shape'], dtype=np.int32), (N, 1))
    for prec in ('fp32', 'f16x3'):
        e = engs[prec]
        out = e.forward(torch.from_numpy(k['img']).to('cuda:0'), k['T'], img_hw=hw)
        got = out['gaze'][0].cpu()
        rep = PT.stage_report(e, prec, sd, k['img'], k['metas'], k['T'], stages)
        assert max(rep['teacher_forced']) < 2e-5, (prec, rep['teacher_forced'])   # measured <= 4.2e-6 (f16x3), 3.2e-6 (fp32)
        ang = 2 * torch.asin(((got.double() - ref['gaze_score'].double()).norm(dim=-1) / 2).clamp(max=1))   # acos(dot) has no resolution near 0
        d = orc.yaw_pitch_diff(got, ref['gaze_score']).max(dim=1).values
        away = ref['gaze_score'][:, 1].abs() < 0.99
        clip_crossed = torch.from_numpy(rep['crossed_boxes'].reshape(k['B'], k['T'] * 3).any(axis=1))     # [B]
        frame_ok = ~clip_crossed.repeat_interleave(k['T'])                                                 # [N]: frames of clips that crossed nothing
        _fuzz_seen[(index, prec)] = bool(clip_crossed.any())
        print(f'fuzz case {index} {prec}: max angle {float(ang.max()):.2e} rad, max d(yaw, pitch) {float(d.max()):.2e}, '
              f'clips that crossed a discontinuity: {int(clip_crossed.sum())} of {k["B"]}; {PT.describe(rep)}')
        assert bool(clip_crossed.any()) == bool(rep['discontinuity'])
        if bool(frame_ok.any()):
            assert float(ang[frame_ok].max()) < 2e-4, (prec, float(ang[frame_ok].max()))      # measured over 2400 inputs: fp32 <= 6.9e-5, f16x3 <= 1.1e-4
            sel = frame_ok & away
            assert not bool(sel.any()) or float(d[sel].max()) < F32_TOL, (prec, float(d[sel].max()))
        if bool((~frame_ok).any()):
            assert float(ang[~frame_ok].max()) < FUZZ_CROSSED_CLIP_BOUND, (prec, float(ang[~frame_ok].max()))


@pytest.mark.parametrize('prec', ['fp32', 'f16x3'])
def test_parity_fuzz_discontinuity_rate(prec):
    """The rate the fuzz cases above are allowed to cross a model discontinuity at: none for fp32, at most one of the 16 for f16x3
    (measured over 2400 inputs: 0 and 1).  Runs after them (file order); a partial selection of the cases asserts on what ran."""
    seen = [v for (i, p), v in _fuzz_seen.items() if p == prec]
    if not seen:
        pytest.skip('the fuzz cases did not run in this session')
    assert sum(seen) <= FUZZ_MAX_CROSSINGS[prec], (prec, sum(seen), len(seen))


@pytest.mark.parametrize('precision', ['f16x3', 'bf16'])
def test_two_threads_two_engines_one_device(precision):
    """include/mcgaze_hip.h "Threading and streams": an engine runs one forward at a time, so concurrent forwards on one device take
    one engine per thread (here: two engines of one precision over the SAME weights, each thread on its own stream, sharing the device's
    side-stream pool).  Eight forwards per thread on different inputs, submitted concurrently, must reproduce the single-threaded
    results bit for bit; so must two threads hammering ONE engine (serialised by its mutex), each with its own workspace."""
    import threading
    from mcgaze_amd.engine import HipEngine
    sd = synth.make_state_dict(0)
    T, B = 7, 10                      # 70 frames: the trunk splits into two concurrent frame ranges
    engines = [HipEngine(sd, precision=precision) for _ in range(2)]
    imgs = [[torch.from_numpy(synth.make_clips(500 + 10 * t + i, B, T)).to('cuda:0') for i in range(4)] for t in range(2)]
    want = [[{k: v.clone() for k, v in engines[0].forward(x, T).items()} for x in imgs[t]] for t in range(2)]
    torch.cuda.synchronize()

    def hammer(use_engines):
        got, errs = [[None] * 8 for _ in range(2)], []

        def work(t):
            try:
                s = torch.cuda.Stream(device='cuda:0')
                with torch.cuda.stream(s):
                    for r in range(8):
                        got[t][r] = {k: v.clone() for k, v in use_engines[t].forward(imgs[t][r % 4], T).items()}
                s.synchronize()
            except Exception as ex:   # surfaced below: a failure inside a thread must fail the test
                errs.append(ex)
        th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
        for t in range(2):
            for r in range(8):
                for k in ('gaze', 'boxes', 'scores'):
                    assert torch.equal(got[t][r][k], want[t][r % 4][k]), (t, r, k)

    hammer(engines)                                   # one engine per thread
    third = HipEngine(sd, precision=precision)          # ONE engine, two threads: HipEngine.forward re-uses its workspace, so give the
    class _OwnWs:                                     # second thread a view of the engine with its own workspace
        def __init__(self, e):
            self.e, self.ws = e, None
        def forward(self, x, T):
            import ctypes as C
            from mcgaze_amd import lib as L
            from mcgaze_amd.engine import _ptr, _stream, _ws
            e = self.e
            N, _, H, W = x.shape
            if self.ws is None:
                self.ws = _ws(e.lib.mcg_engine_workspace_bytes(e._handle, N, H, W, 0), e.device)
            out = dict(gaze=torch.empty(4, N, 3, device=e.device), boxes=torch.empty(N, 3, 4, device=e.device), scores=torch.empty(N, 3, device=e.device))
            L.check(e.lib.mcg_clip_forward(e._handle, _stream(e.device), _ptr(x), N, T, H, W, C.c_void_p(0), 0, _ptr(out['gaze']), _ptr(out['boxes']),
                                           _ptr(out['scores']), _ptr(self.ws), self.ws.numel()), 'mcg_clip_forward')
            return out
    hammer([_OwnWs(third), _OwnWs(third)])


def test_range_audit_counts_values_beyond_the_fp16_range():
    """the debug option `range_audit` tallies, per activation tensor of the trunk, the values an f16x3 operand half
    cannot hold (|x| > 65504) and the non-finite ones.  The synthetic net is clean (all zeros); the same net with its stem conv scaled by
    1e6 trips the counter at the first tensor and -- the halves saturate instead of overflowing -- stays finite after it."""
    from mcgaze_amd.engine import HipEngine
    sd = synth.make_state_dict(0)
    img = torch.from_numpy(synth.make_clips(3, 2, 7)).to('cuda:0')
    e = HipEngine(sd, precision='f16x3')
    e.set_option('range_audit', 1)
    e.forward(img, 7)
    rec = e.range_audit()
    names = [r[0] for r in rec]
    assert names[0] == 'stem' and 'fpn.P2' in names and any(n.startswith('layer3.5') for n in names) and len(rec) > 40
    assert all(big == 0 and bad == 0 for _, big, bad in rec), [r for r in rec if r[1] or r[2]][:4]
    hot = dict(sd)
    hot['backbone.conv1.weight'] = sd['backbone.conv1.weight'] * 1e6
    e2 = HipEngine(hot, precision='f16x3')
    e2.set_option('range_audit', 1)
    out = e2.forward(img, 7)
    rec2 = e2.range_audit()
    assert rec2[0][0] == 'stem' and rec2[0][1] > 1000 and rec2[0][2] == 0, rec2[:3]
    assert sum(r[2] for r in rec2) == 0 and bool(torch.isfinite(out['gaze']).all())   # saturating halves: large, never inf / nan
    assert all(r[1] == 0 and r[2] == 0 for r in e2.range_audit())                     # the read resets the counters
    e2.set_option('range_audit', 0)
    with pytest.raises(Exception, match='range_audit option is off'):
        e2.range_audit()
