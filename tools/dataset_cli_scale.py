"""GPU tool: BASELINE.json configs[4]'s command line at the SIZE of the Gaze360 test split, on synthetic frames (no dataset, no checkpoint in
this image): a random-init checkpoint file + an annotation json + JPEG frames on disk -> `tools/test_gaze360_gaze.py <config> <checkpoint> --json ...
--root ... --anno ...` run as the user would run it (a child process, wall clock around the whole command: interpreter start, checkpoint
ingestion and weight packing, decode, device preprocessing, windows, engine, merge, result file, MAE), then `tools/calculate_mae_gaze360.py`
on the result file.  usage: python tools/dataset_cli_scale.py [videos=433] [frames_per_video=60] [side=360] [precision=f16x3]
(433 x 60 = 25 980 frames; the Gaze360 test split the reference quotes its MAE on has 25 969 annotated frames.)"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from mcgaze_amd import synth  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 433
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60
S = int(sys.argv[3]) if len(sys.argv) > 3 else 360
prec = sys.argv[4] if len(sys.argv) > 4 else 'f16x3'
rs = np.random.RandomState(0)
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.time()
    ckpt = os.path.join(tmp, 'ckpt.pth')
    torch.save(dict(meta=dict(CLASSES=('face', 'eyes', 'head')), state_dict={'module.' + k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(0).items()}), ckpt)
    base = rs.randint(0, 256, (S // 8, S // 8, 3)).astype(np.uint8)
    videos, annos = [], []
    for v in range(V):
        os.makedirs(os.path.join(tmp, 'frames', f'v{v}'))
        names = []
        for i in range(L):   # smooth content (JPEG-like statistics), different per frame
            img = np.asarray(Image.fromarray(np.roll(base, (v + i) % 17, axis=1)).resize((S, S), Image.BILINEAR))
            names.append(f'v{v}/{i:06d}.jpg')
            Image.fromarray(img).save(os.path.join(tmp, 'frames', names[-1]), quality=90)
        videos.append(dict(id=v + 1, file_names=names))
        g = rs.randn(L, 3)
        annos.append(dict(gaze=(g / np.linalg.norm(g, axis=1, keepdims=True)).tolist()))
    test_json = os.path.join(tmp, 'test.json')
    json.dump(dict(videos=videos, annotations=annos), open(test_json, 'w'))
    print(f'{V} videos x {L} frames of {S}x{S} JPEG = {V * L} frames written in {time.time() - t0:.1f} s; engine {prec}', flush=True)
    cfg = os.path.join(root, 'configs', 'mcgaze', 'r50_clip7_gaze360.py')
    for rep in range(2):                 # the second run has the frames in the page cache, like a second evaluation of the same split
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'test_gaze360_gaze.py'), cfg, ckpt, '--json', test_json, '--root', os.path.join(tmp, 'frames'),
                            '--seed', '3', '--anno', test_json, '--precision', prec], cwd=tmp, capture_output=True, text=True)
        dt = time.time() - t0
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-4000:])
            raise SystemExit(f'the dataset tool failed ({r.returncode})')
        print(f'run {rep}: tools/test_gaze360_gaze.py wall clock {dt:.2f} s = {V * L / dt:.0f} frames/s for the whole command; its output:', flush=True)
        print('\n'.join('    ' + ln for ln in (r.stdout.strip().splitlines() + [ln for ln in r.stderr.splitlines() if ln.startswith('[mcgaze_amd]')])), flush=True)
    res = os.path.join(tmp, 'results', 'results_r50_clip7_gaze360_test.json')
    recs = json.load(open(res))
    assert [r_['video_id'] for r_ in recs] == [v['id'] for v in videos] and all(len(r_['fusion_gazes']) == L for r_ in recs)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'calculate_mae_gaze360.py'), '--evalfile', res, '--anno', test_json], cwd=tmp, capture_output=True, text=True)
    print(f'tools/calculate_mae_gaze360.py on the result file ({os.path.getsize(res) / 1e6:.1f} MB, {len(recs)} records): {time.time() - t0:.2f} s, rc {r.returncode}')
    print('\n'.join('    ' + ln for ln in (r.stdout + r.stderr).strip().splitlines()[-8:]))
