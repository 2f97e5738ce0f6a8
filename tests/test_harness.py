"""windowing + overlap merge and the MAE metric against goldens produced by running
the reference's own tools/test_gaze360_gaze.py::main (with a seeded fake model) and tools/calculate_mae_*.py
(oracle/dev/make_harness_goldens.py)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from mcgaze_amd import harness, metric, synth


@pytest.fixture(scope='module')
def merge_golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'harness_merge.json')))


def test_windows_match_the_reference_harness(merge_golden):
    calls = merge_golden['calls']
    for vid, L in enumerate(merge_golden['video_lengths'], start=1):
        want = [c['frames'] for c in calls if c['video'] == vid]
        got = [list(range(a, b)) for a, b, _ in harness.plan_windows(L)]
        assert got == want, (L, got, want)
    # closed-form facts (tools/test_gaze360_gaze.py:72-86)
    assert harness.plan_windows(7) == [(0, 7, 0)] and harness.plan_windows(3) == [(0, 3, 0)]
    assert harness.plan_windows(8) == [(0, 7, 3), (1, 8, 6)]
    assert harness.plan_windows(15) == [(0, 7, 3), (4, 11, 3), (8, 15, 3)]


def test_merge_reproduces_the_reference_results(merge_golden):
    calls = merge_golden['calls']
    idx = 0
    for vid, L in enumerate(merge_golden['video_lengths'], start=1):
        plan = harness.plan_windows(L)
        outs = []
        for (a, b, _) in plan:
            assert calls[idx]['video'] == vid
            outs.append(synth.fake_clip_outputs(vid, list(range(a, b)), call_index=idx))
            idx += 1
        rec = harness.video_record(vid, *harness.merge_video(plan, outs))
        want = merge_golden['results'][vid - 1]
        assert rec['video_id'] == want['video_id'] and rec['category_id'] == 1 and set(rec) == set(want)
        for k, v in want.items():
            if k.endswith('_bboxes'):
                assert [x is None for x in rec[k]] == [x is None for x in v], (vid, k)
                np.testing.assert_allclose([x for x in rec[k] if x is not None], [x for x in v if x is not None], atol=1e-5)
            elif isinstance(v, list):
                np.testing.assert_allclose(np.array(rec[k], dtype=np.float64), np.array(v, dtype=np.float64), atol=1e-6, err_msg=f'{vid} {k}')
        assert len(rec['fusion_gazes']) == L
    assert idx == len(calls)
    assert harness.result_file_name('configs/x/fake_cfg.py', '/tmp/abc/fake_test.json') == merge_golden['result_file_name']


def test_metric_matches_the_reference_scripts(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'metric_kat.json')))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        metric.gaze_error(g['eval'], g['anno'], 'fusion_gazes')
        metric.gaze_error(g['eval'], g['anno'], 'face_gazes')
    assert buf.getvalue() == g['gaze360_printed']
    anno3 = dict(annotations=[a for a in g['anno']['annotations'] for _ in range(3)])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        metric.gaze_error(g['eval'], anno3, 'fusion_gazes', setting='l2cs')
    assert buf.getvalue() == g['l2cs_printed']
    k = g['kat']
    gt, p = torch.tensor(k['g']), torch.tensor(k['p'])
    sm = metric.smooth_filter(p.clone())
    np.testing.assert_allclose(sm.numpy(), np.array(k['smooth']), atol=1e-6)
    # SURVEY.md section 8(f)-2 [probe] values
    np.testing.assert_allclose(sm.numpy(), [[.2037, .0967, -.9742], [.2076, .0983, -.9733], [.1428, .0341, -.9892]], atol=1e-4)
    assert abs(float(metric.compute_angular_error(sm, gt)) - 6.7325) < 1e-3 and abs(float(metric.compute_yaw_angular(gt[0])) - 5.7106) < 1e-3
    assert abs(float(metric.compute_angular_error(gt, gt))) < 0.05  # pred == gt -> ~0 (acos rounding only)
    assert abs(float(metric.compute_pitch_angular(gt[0])) - k['pitch0']) < 1e-5


@pytest.mark.gpu
def test_run_videos_equals_window_by_window(golden_dir):
    """Batched driver == one engine call per window + merge (bitwise), over videos of mixed length incl. L < 7."""
    from mcgaze_amd.engine import HipEngine
    e = HipEngine(synth.make_state_dict(0), precision='bf16')
    lengths = [3, 7, 9, 16]
    videos = [dict(id=i + 1, frames=torch.from_numpy(synth.make_clips(50 + i, 1, L))) for i, L in enumerate(lengths)]
    recs = harness.run_videos(e, videos, batch_clips=4, scale_factor=(1.4, 1.4, 1.4, 1.4))
    for v, rec in zip(videos, recs):
        plan = harness.plan_windows(v['frames'].shape[0])
        outs = []
        for a, b, _ in plan:
            o = e.forward(v['frames'][a:b].to('cuda:0').contiguous(), b - a)
            det = torch.cat([o['boxes'] / torch.tensor([1.4] * 4, device='cuda:0'), o['scores'][..., None]], dim=-1)
            outs.append((det.clone(), o['gaze'][0].clone(), o['gaze'][1:].permute(1, 0, 2).clone()))
        want = harness.video_record(v['id'], *harness.merge_video(plan, outs))
        assert rec == want
        assert len(rec['fusion_gazes']) == v['frames'].shape[0]


@pytest.mark.gpu
def test_run_annotation_from_files_equals_window_by_window(tmp_path):
    """Annotation file -> decoded frames -> device preprocessing (random crop per window, seeded) -> batched engine ->
    merge, against the same thing done one window at a time (tools/test_gaze360_gaze.py:57-269 end to end)."""
    from PIL import Image
    from mcgaze_amd import Config
    from mcgaze_amd.engine import HipEngine
    from mcgaze_amd.pipeline import DevicePipeline
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pipe = DevicePipeline(Config.fromfile(os.path.join(root, 'configs', 'mcgaze', 'r50_clip7_gaze360.py')).data.test.pipeline)
    e = HipEngine(synth.make_state_dict(0), precision='bf16')
    rs = np.random.RandomState(4)
    anno = dict(videos=[])
    for vid, (L, shape) in enumerate([(9, (250, 250, 3)), (5, (300, 280, 3)), (12, (250, 250, 3))]):
        names = []
        for i in range(L):
            names.append(f'v{vid}/{i:06d}.png')
            os.makedirs(str(tmp_path / f'v{vid}'), exist_ok=True)
            Image.fromarray(rs.randint(0, 256, shape).astype(np.uint8)).save(str(tmp_path / names[-1]))
        anno['videos'].append(dict(id=vid + 10, file_names=names))
    recs = harness.run_annotation(e, anno, str(tmp_path), pipe, batch_clips=3, rng=np.random.RandomState(21))   # in-line decode, cached
    for workers, lookahead in ((8, None), (3, 1)):   # decode threads ahead of the GPU / a short look-ahead with evictions: the records may not depend on it
        assert harness.run_annotation(e, anno, str(tmp_path), pipe, batch_clips=3, rng=np.random.RandomState(21), workers=workers, lookahead=lookahead) == recs
    for workers, lookahead in ((4, None), (2, 1)):   # the same with decode helper PROCESSES (shared-memory ring)
        assert harness.run_annotation(e, anno, str(tmp_path), pipe, batch_clips=3, rng=np.random.RandomState(21), workers=workers, lookahead=lookahead,
                                      processes=True) == recs
    rng = np.random.RandomState(21)
    for v, rec in zip(anno['videos'], recs):
        plan = harness.plan_windows(len(v['file_names']))
        outs = []
        for a, b, _ in plan:
            img, metas = pipe(v['file_names'][a:b], device='cuda:0', rng=rng, img_prefix=str(tmp_path))
            o = e.forward(img, b - a, img_hw=[m['img_shape'][:2] for m in metas])
            sc = torch.as_tensor(np.stack([m['scale_factor'] for m in metas])).to('cuda:0')[:, None, :]
            det = torch.cat([o['boxes'] / sc, o['scores'][..., None]], dim=-1)
            outs.append((det.clone(), o['gaze'][0].clone(), o['gaze'][1:].permute(1, 0, 2).clone()))
        want = harness.video_record(v['id'], *harness.merge_video(plan, outs))
        assert rec == want and len(rec['fusion_gazes']) == len(v['file_names'])


def test_frame_cache_helper_processes(tmp_path):
    """pipeline.FrameCache(processes=True): frames decoded by the helper processes (mcgaze_amd/_decode_worker.py, shared-memory ring) are
    the in-line decoder's pixels; a frame larger than a ring slot is decoded in line; a missing file raises where the frame is asked
    for; eviction under a small capacity hands slots out again without mixing frames up."""
    from PIL import Image
    from mcgaze_amd.pipeline import FrameCache, LoadImageFromFile
    rs = np.random.RandomState(3)
    paths = []
    for i, shape in enumerate([(40, 50, 3), (64, 64, 3), (33, 70, 3), (40, 50), (90, 120, 3), (40, 50, 3), (12, 12, 3), (50, 40, 3)]):
        paths.append(str(tmp_path / f'{i}.png'))
        Image.fromarray(rs.randint(0, 256, shape).astype(np.uint8)).save(paths[-1])      # index 3: a grey image (converted to RGB by both decoders)
    want = [LoadImageFromFile.load(p, rgb=True) for p in paths]
    cache = FrameCache(workers=3, capacity=4, processes=True, slot_bytes=64 * 64 * 3)      # the 90x120 frame does not fit a slot
    try:
        for rounds in range(2):
            cache.prefetch(paths[:4])
            for k in (0, 1, 2, 3, 4, 5, 6, 7, 2, 0):
                got = cache(paths[k])
                assert got.dtype == np.uint8 and got.shape == want[k].shape and np.array_equal(got, want[k]), k
                cache.release()                                  # done with the view: its slot may be evicted again
        assert cache.decodes >= len(paths)
        with pytest.raises(RuntimeError, match='decode worker'):
            cache(str(tmp_path / 'absent.png'))
    finally:
        cache.close()


def test_frame_cache_never_recycles_a_view_that_is_still_out(tmp_path):
    """a ring view handed out by FrameCache(processes=True) stays valid until release() -- a later miss must not evict its slot
    and let another frame's pixels show through the view.  More views out than the cache has slots raises instead of corrupting, and
    DevicePipeline.run_many's contract (release once the staging copy is made) is what makes a small cache usable across calls."""
    from PIL import Image
    from mcgaze_amd.pipeline import FrameCache, LoadImageFromFile
    rs = np.random.RandomState(5)
    paths = []
    for i in range(8):
        paths.append(str(tmp_path / f'{i}.png'))
        Image.fromarray(rs.randint(0, 256, (24, 20, 3)).astype(np.uint8)).save(paths[-1])
    want = [LoadImageFromFile.load(p, rgb=True) for p in paths]
    cache = FrameCache(workers=2, capacity=4, processes=True, slot_bytes=24 * 20 * 3)
    try:
        views = [cache(p) for p in paths[:4]]
        with pytest.raises(RuntimeError, match='handed out as views'):
            cache(paths[4])                                      # a fifth view needs a slot; every slot backs a view that is still out
        for v, w in zip(views, want):
            assert np.array_equal(v, w)                          # ... and the four that are out are untouched
        cache.release()
        for k in range(4, 8):                                    # after release the slots go round again
            assert np.array_equal(cache(paths[k]), want[k])
        cache.release()
        assert np.array_equal(cache(paths[0]), want[0])
    finally:
        cache.close()


def test_frame_cache_falls_back_to_inline_when_shm_cannot_back_the_ring(tmp_path, monkeypatch):
    """the decode ring is a sparse /dev/shm file; a tmpfs that runs out under a mapped write is a SIGBUS in a helper, not
    an exception (a container's default /dev/shm is 64 MB).  FrameCache checks the free space -- this rank's share of it (LOCAL_WORLD_SIZE)
    -- against the ring priced at the caller's expected frame size (else at whole slots) and decodes IN LINE, with a warning, when it does
    not fit -- same pixels, no helper processes, no threads."""
    import os as _os
    from PIL import Image
    from mcgaze_amd import pipeline as P
    rs = np.random.RandomState(9)
    paths = []
    for i in range(3):
        paths.append(str(tmp_path / f'{i}.png'))
        Image.fromarray(rs.randint(0, 256, (20, 16, 3)).astype(np.uint8)).save(paths[-1])
    real = _os.statvfs

    class Tiny:
        def __init__(self, st):
            self.f_frsize, self.f_bavail = st.f_frsize, (8 << 20) // st.f_frsize        # 8 MiB free
    monkeypatch.setattr(P.os, 'statvfs', lambda p: Tiny(real(p)))
    with pytest.warns(UserWarning, match='decoding in line'):
        cache = P.FrameCache(workers=2, capacity=64, processes=True)               # 65 slots x 3 MiB >> 8 MiB
    try:
        assert cache.procs is None and cache.pool is None
        cache.prefetch(paths)
        for pth in paths:
            assert np.array_equal(cache(pth), P.LoadImageFromFile.load(pth, rgb=True))
    finally:
        cache.close()
    # priced at the frames the caller expects (20 x 16 x 3 bytes -> one page per slot) the same ring fits the same 8 MiB ...
    cache = P.FrameCache(workers=2, capacity=64, processes=True, frame_bytes=20 * 16 * 3)
    try:
        assert cache.procs is not None
        cache.prefetch(paths)
        for pth in paths:
            assert np.array_equal(cache(pth), P.LoadImageFromFile.load(pth, rgb=True))
        cache.release()
    finally:
        cache.close()
    # ... but not when 64 ranks of the node share the tmpfs (0.8 x 8 MiB / 64 = 25 pages each)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '64')
    with pytest.warns(UserWarning, match='64 rank'):
        cache = P.FrameCache(workers=2, capacity=64, processes=True, frame_bytes=20 * 16 * 3)
    assert cache.procs is None and cache.pool is None
    cache.close()


def test_dataset_tool_cli_and_sharding():
    """tools/test_gaze360_gaze.py keeps the reference's command line (its :20-44) and shards whole videos by frame count."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('dataset_tool', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'test_gaze360_gaze.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    a = tool.parse_args(['cfg.py', 'ckpt.pth'])
    assert (a.json, a.root, a.device) == ('data/gaze360/test.json', 'data/gaze360/test_rawframes/', 'cuda:0')   # the reference's defaults
    a = tool.parse_args(['cfg.py', 'ckpt.pth', '--json', 'x.json', '--cfg-options', 'model.test_cfg.a=3', 'clip_length=5', 'k=1,2', 'f=true'])
    assert a.cfg_options == {'model.test_cfg.a': 3, 'clip_length': 5, 'k': [1, 2], 'f': True}
    videos = [dict(file_names=['f'] * n) for n in (50, 7, 7, 30, 12, 40, 3)]
    shards = [tool.shard_videos(videos, 3, r) for r in range(3)]
    assert sorted(i for s in shards for i in s) == list(range(7))
    loads = [sum(len(videos[i]['file_names']) for i in s) for s in shards]
    assert max(loads) - min(loads) <= 12 and shards == [sorted(s) for s in shards]


@pytest.mark.gpu
def test_dataset_tool_end_to_end(tmp_path, monkeypatch, capsys):
    """Checkpoint file + annotation json + PNG frames -> results_<cfg>_<json> with the reference's schema, MAE printed."""
    import importlib.util
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('dataset_tool', os.path.join(root, 'tools', 'test_gaze360_gaze.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    rs = np.random.RandomState(6)
    ckpt = str(tmp_path / 'ckpt.pth')
    torch.save(dict(meta=dict(CLASSES=('face', 'eyes', 'head')),
                    state_dict={'module.' + k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(0).items()}), ckpt)
    videos, annos = [], []
    for vid, L in enumerate((8, 4)):
        names = []
        os.makedirs(str(tmp_path / 'frames' / f'v{vid}'))
        for i in range(L):
            names.append(f'v{vid}/{i:06d}.png')
            Image.fromarray(rs.randint(0, 256, (240, 240, 3)).astype(np.uint8)).save(str(tmp_path / 'frames' / names[-1]))
        videos.append(dict(id=vid + 1, file_names=names))
        g = rs.randn(L, 3)
        annos.append(dict(gaze=(g / np.linalg.norm(g, axis=1, keepdims=True)).tolist()))
    test_json = str(tmp_path / 'test.json')
    json.dump(dict(videos=videos, annotations=annos), open(test_json, 'w'))
    monkeypatch.chdir(tmp_path)
    cfg = os.path.join(root, 'configs', 'mcgaze', 'r50_clip7_gaze360.py')
    tool.main([cfg, ckpt, '--json', test_json, '--root', str(tmp_path / 'frames'), '--seed', '3', '--anno', test_json, '--batch-clips', '2'])
    out = capsys.readouterr().out
    path = tmp_path / 'results' / 'results_r50_clip7_gaze360_test.json'
    assert path.exists() and 'fusion_gazes mean angular error 360:' in out
    recs = json.load(open(str(path)))
    assert [r['video_id'] for r in recs] == [1, 2] and [len(r['fusion_gazes']) for r in recs] == [8, 4]
    assert set(recs[0]) >= {'video_id', 'category_id', 'fusion_gazes', 'face_bboxes', 'face_gazes', 'face_score', 'eyes_gazes', 'head_gazes'}


@pytest.mark.gpu
def test_dataset_run_matches_the_cpu_oracle_end_to_end(tmp_path):
    """BASELINE.json configs[4] in miniature: frames on disk -> decode -> preprocessing (seeded random crop per window) -> windows ->
    default engine (f16x3) -> overlap merge -> MAE, against the same chain on the CPU: oracle preprocessing + oracle forward per
    window (the reference's arithmetic, pinned by the goldens) + the same merge.  Gaze vectors within north_star's 1e-3 rad, the
    three MAE figures within 0.02 degrees (they are printed with two decimals)."""
    from PIL import Image
    from mcgaze_amd import Config, metric
    from mcgaze_amd.engine import HipEngine
    from mcgaze_amd.pipeline import DevicePipeline
    from oracle import mcgaze_oracle as orc
    from oracle import preprocess_oracle as po
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pipe = DevicePipeline(Config.fromfile(os.path.join(root, 'configs', 'mcgaze', 'r50_clip7_gaze360.py')).data.test.pipeline)
    sd = synth.make_state_dict(0)
    rs = np.random.RandomState(12)
    anno, frames = dict(videos=[], annotations=[]), {}
    for vid, (L, shape) in enumerate([(10, (260, 250, 3)), (5, (300, 280, 3))]):
        names = []
        os.makedirs(str(tmp_path / f'v{vid}'))
        base = rs.randint(0, 256, (shape[0] // 6 + 1, shape[1] // 6 + 1, 3)).astype(np.uint8)
        for i in range(L):   # smooth, frame-dependent content
            img = np.asarray(Image.fromarray(np.roll(base, i, axis=1)).resize((shape[1], shape[0]), Image.BILINEAR))
            names.append(f'v{vid}/{i:06d}.png')
            Image.fromarray(img).save(str(tmp_path / names[-1]))
            frames[names[-1]] = np.ascontiguousarray(img[..., ::-1])        # what cv2.imread would hand the reference: BGR
        g = rs.randn(L, 3)
        anno['videos'].append(dict(id=vid, file_names=names))
        anno['annotations'].append(dict(gaze=(g / np.linalg.norm(g, axis=1, keepdims=True)).tolist()))
    eng = HipEngine(sd, precision='f16x3')
    recs = harness.run_annotation(eng, anno, str(tmp_path), pipe, rng=np.random.RandomState(5))
    # ---- the CPU chain
    rng = np.random.RandomState(5)
    want = []
    for v in anno['videos']:
        plan = harness.plan_windows(len(v['file_names']))
        outs = []
        for a, b, _ in plan:
            chw, metas = [], []
            for n in sorted(v['file_names'][a:b]):
                u = rng.rand(1)[0]          # CenterCrop's draw, then RandomFlip's (transforms.py:1126-1130, 463-497)
                rng.random_sample()
                c, m = po.test_pipeline(frames[n], u=u)
                chw.append(c)
                metas.append(m)
            det, gaze = orc.forward(sd, po.collate_clip(chw), metas, b - a, rescale=True)
            others = torch.stack([gaze['face_gaze_score'], gaze['eyes_gaze_score'], gaze['head_gaze_score']], dim=1)
            outs.append((det, gaze['gaze_score'], others))
        want.append(harness.video_record(v['id'], *harness.merge_video(plan, outs)))
    for rec, ref in zip(recs, want):
        got, exp = torch.tensor(rec['fusion_gazes']), torch.tensor(ref['fusion_gazes'])
        assert got.shape == exp.shape
        assert float(orc.yaw_pitch_diff(got, exp).max()) < 1e-3
        for c in ('face', 'eyes', 'head'):
            assert float((torch.tensor(rec[f'{c}_gazes']) - torch.tensor(ref[f'{c}_gazes'])).abs().max()) < 1e-3
            assert np.allclose(np.array(rec[f'{c}_score']), np.array(ref[f'{c}_score']), atol=1e-3)
    a = metric.gaze_error(recs, anno, verbose=False)
    b = metric.gaze_error(want, anno, verbose=False)
    for k in a:
        assert abs(a[k] - b[k]) < 0.02 or (a[k] != a[k] and b[k] != b[k]), (k, a[k], b[k])


def test_mae_command_lines_print_the_reference_lines(golden_dir, tmp_path):
    """tools/calculate_mae_{gaze360,l2cs}.py keep the reference scripts' flags (--evalfile / --anno) and printed lines."""
    import subprocess, sys
    g = json.load(open(os.path.join(golden_dir, 'metric_kat.json')))
    ev, an, an3 = tmp_path / 'eval.json', tmp_path / 'anno.json', tmp_path / 'anno3.json'
    ev.write_text(json.dumps(g['eval']))
    an.write_text(json.dumps(g['anno']))
    an3.write_text(json.dumps(dict(annotations=[a for a in g['anno']['annotations'] for _ in range(3)])))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'calculate_mae_gaze360.py'), '--evalfile', str(ev), '--anno', str(an)],
                         capture_output=True, text=True, check=True).stdout
    assert out == g['gaze360_printed'].split('face_gazes')[0]
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'calculate_mae_l2cs.py'), '--evalfile', str(ev), '--anno', str(an3)],
                         capture_output=True, text=True, check=True).stdout
    assert out == g['l2cs_printed']


def test_frame_cache_decodes_each_file_once_ahead_of_the_consumer():
    """pipeline.FrameCache: decode on host threads, one decode per file while it stays cached, LRU eviction, loader errors surface
    at the consumer."""
    import threading, time
    from mcgaze_amd.pipeline import FrameCache
    seen, lock = [], threading.Lock()

    def loader(path):
        if path == 'bad':
            raise OSError('cannot decode')
        time.sleep(0.01)
        with lock:
            seen.append(path)
        return np.full((2, 2, 3), int(path), dtype=np.uint8)

    c = FrameCache(workers=0, capacity=2, loader=loader)                           # in line: prefetch is a no-op, decode on demand, LRU
    c.prefetch(['1', '2'])
    assert c.decodes == 0 and [int(c(p)[0, 0, 0]) for p in ('1', '2', '1', '3', '1', '2')] == [1, 2, 1, 3, 1, 2] and seen == ['1', '2', '3', '2']
    c.close()
    seen.clear()
    c = FrameCache(workers=4, capacity=6, loader=loader)
    c.prefetch(str(i) for i in range(6))
    assert [int(c(str(i))[0, 0, 0]) for i in (0, 1, 2, 3, 4, 5, 2, 3)] == [0, 1, 2, 3, 4, 5, 2, 3]
    assert sorted(seen) == [str(i) for i in range(6)] and c.decodes == 6          # overlapping windows: no second decode
    c.prefetch(['6', '7'])                                                         # capacity 6: the least recently used (0, 1) go
    assert int(c('7')[0, 0, 0]) == 7 and int(c('0')[0, 0, 0]) == 0 and c.decodes == 9
    c.prefetch(['bad'])
    with pytest.raises(OSError):
        c('bad')
    c.close()
