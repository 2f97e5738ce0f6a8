"""DEV-ONLY: goldens for SURVEY.md section 8(f)-1/2 (clip windowing + overlap merge, MAE metric).

Runs the reference's OWN code in the build container:
  * tools/test_gaze360_gaze.py::main  -- executed unchanged (imported through the mmcv stand-in) with the model,
    the frame pipeline and collate/scatter replaced by deterministic fakes, so that exactly the windowing and
    merging logic of lines 60-269 runs on seeded per-clip outputs;
  * tools/calculate_mae_gaze360.py / calculate_mae_l2cs.py -- imported as they are (pure torch).
Writes tests/golden/harness_merge.json and tests/golden/metric_kat.json (inputs by seed + expected outputs).
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import tempfile
import types
import warnings

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import mmcv_standin  # noqa: E402,F401

sys.path.insert(0, '/root/reference')
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mcgaze_amd.synth import fake_clip_outputs  # noqa: E402  (the seeded fake model shared with the tests)

OUT = os.path.join(ROOT, 'tests', 'golden')
VIDEO_LENGTHS = [3, 7, 8, 10, 11, 15, 20, 33, 1]


def load_module(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _DC:  # stands for mmcv DataContainer: only .data is used (test_gaze360_gaze.py:96-100)
    def __init__(self, data):
        self.data = data


class FakeModel:
    """Returns the seeded fake outputs of mcgaze_amd.harness.fake_clip_outputs for the frames it is shown."""

    def __init__(self):
        self.cfg = types.SimpleNamespace(data=types.SimpleNamespace(test=types.SimpleNamespace(pipeline=[])))
        self.calls = []

    def start_flops_count(self):
        pass

    def stop_flops_count(self):
        pass

    def __call__(self, return_loss, rescale, format, img, img_metas):
        assert return_loss is False and rescale is True and format is False
        names = [m['filename'] for m in img_metas[0]]
        vid = int(names[0].split('/')[0][1:])
        frames = [int(n.split('/')[1].split('.')[0]) for n in names]
        det, fused, others = fake_clip_outputs(vid, frames, call_index=len(self.calls))
        self.calls.append((vid, frames))
        gaze = dict(gaze_score=fused, face_gaze_score=others[:, 0], eyes_gaze_score=others[:, 1], head_gaze_score=others[:, 2])
        return ([det[i] for i in range(det.shape[0])], [[0, 1, 2]] * det.shape[0]), gaze


def merge_goldens():
    harness = load_module('/root/reference/tools/test_gaze360_gaze.py', 'ref_harness')
    model = FakeModel()
    harness.init_detector = lambda *a, **k: model
    harness.add_flops_counting_methods = lambda m: m
    harness.Compose = lambda cfg: (lambda data: dict(img_metas=_DC(dict(filename=data['img_info']['filename'])), img=_DC(torch.zeros(1))))
    harness.collate = lambda datas, samples_per_gpu: dict(img_metas=_DC([[d['img_metas'].data for d in datas]]), img=_DC([torch.zeros(len(datas))]))
    harness.scatter = lambda datas, devices: [datas]
    videos = [dict(id=i + 1, file_names=[f'v{i + 1}/{f:06d}.png' for f in range(L)]) for i, L in enumerate(VIDEO_LENGTHS)]
    with tempfile.TemporaryDirectory() as tmp:
        anno = os.path.join(tmp, 'fake_test.json')
        json.dump(dict(videos=videos), open(anno, 'w'))
        args = types.SimpleNamespace(config='configs/x/fake_cfg.py', checkpoint=None, json=anno, root='', device='cpu', cfg_options=None)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                harness.main(args)
            result_file = os.path.join(tmp, 'results', 'results_fake_cfg_fake_test.json')
            results = json.load(open(result_file))
        finally:
            os.chdir(cwd)
    golden = dict(video_lengths=VIDEO_LENGTHS, calls=[dict(video=v, frames=f) for v, f in model.calls], results=results,
                  result_file_name='results_fake_cfg_fake_test.json')
    json.dump(golden, open(os.path.join(OUT, 'harness_merge.json'), 'w'))
    print('merge golden:', len(results), 'videos,', len(model.calls), 'model calls; windows of video 4 (L=10):',
          [c[1] for c in model.calls if c[0] == 4])


def metric_goldens():
    g360 = load_module('/root/reference/tools/calculate_mae_gaze360.py', 'ref_mae_gaze360')
    l2cs = load_module('/root/reference/tools/calculate_mae_l2cs.py', 'ref_mae_l2cs')
    rs = np.random.RandomState(7)
    videos, annos = [], []
    for vid, L in enumerate([1, 2, 5, 9, 30]):
        gt = rs.standard_normal((L, 3)).astype(np.float32)
        gt[:, 2] -= 0.8  # mostly frontal (z < 0) with some back-facing frames
        pred = gt / np.linalg.norm(gt, axis=1, keepdims=True) + 0.15 * rs.standard_normal((L, 3)).astype(np.float32)
        pred /= np.linalg.norm(pred, axis=1, keepdims=True)
        pred[::4] *= 0.9  # un-normalised predictions, as the overlap averaging produces
        videos.append(dict(video_id=vid + 1, fusion_gazes=pred.tolist(), face_gazes=(pred[::-1]).tolist()))
        annos.append(dict(gaze=gt.tolist()))
    out = dict(eval=videos, anno=dict(annotations=annos))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g360.gaze_error(videos, dict(annotations=annos), 'fusion_gazes')
        g360.gaze_error(videos, dict(annotations=annos), 'face_gazes')
    out['gaze360_printed'] = buf.getvalue()
    # l2cs reads annotation index anno_id * 3 (calculate_mae_l2cs.py:110)
    annos3 = [a for a in annos for _ in range(3)]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        l2cs.gaze_error(videos, dict(annotations=annos3), 'fusion_gazes')
    out['l2cs_printed'] = buf.getvalue()
    g = torch.tensor([[.1, 0, -1], [.2, .1, -.9], [0, -.1, -1]])
    g = g / g.norm(dim=1, keepdim=True)
    p = (g + 0.05)
    p = p / p.norm(dim=1, keepdim=True)
    out['kat'] = dict(g=g.tolist(), p=p.tolist(), smooth=g360.smooth_filter(p.clone()).tolist(),
                      err=float(g360.compute_angular_error(g360.smooth_filter(p.clone()), g)),
                      yaw0=float(g360.compute_yaw_angular(g[0])), pitch0=float(l2cs.compute_pitch_angular(g[0])))
    json.dump(out, open(os.path.join(OUT, 'metric_kat.json'), 'w'))
    print('metric golden:\n' + out['gaze360_printed'] + out['l2cs_printed'], out['kat']['err'], out['kat']['yaw0'])


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    merge_goldens()
    metric_goldens()
