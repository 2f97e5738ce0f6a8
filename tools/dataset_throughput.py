"""GPU tool: end-to-end rate of the dataset path (files -> decode -> device preprocessing -> windows -> engine -> merged records,
harness.run_annotation) on a synthetic directory of JPEG frames, with the frames decoded in line and by the look-ahead thread pool.
usage: python tools/dataset_throughput.py [videos=160] [frames_per_video=60] [side=360] [precision=f16x3]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from mcgaze_amd import Config, harness, synth
from mcgaze_amd.engine import HipEngine
from mcgaze_amd.pipeline import DevicePipeline

V = int(sys.argv[1]) if len(sys.argv) > 1 else 160
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60
S = int(sys.argv[3]) if len(sys.argv) > 3 else 360
prec = sys.argv[4] if len(sys.argv) > 4 else 'f16x3'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pipe = DevicePipeline(Config.fromfile(os.path.join(root, 'configs', 'mcgaze', 'r50_clip7_gaze360.py')).data.test.pipeline)
eng = HipEngine(synth.make_state_dict(0), precision=prec)
rs = np.random.RandomState(0)
with tempfile.TemporaryDirectory() as tmp:
    anno = dict(videos=[])
    base = rs.randint(0, 256, (S // 8, S // 8, 3)).astype(np.uint8)
    for v in range(V):
        os.makedirs(os.path.join(tmp, f'v{v}'))
        names = []
        for i in range(L):   # smooth content (JPEG-like statistics), different per frame
            img = np.asarray(Image.fromarray(np.roll(base, (v + i) % 17, axis=1)).resize((S, S), Image.BILINEAR))
            names.append(f'v{v}/{i:06d}.jpg')
            Image.fromarray(img).save(os.path.join(tmp, names[-1]), quality=90)
        anno['videos'].append(dict(id=v, file_names=names))
    nwin = sum(len(harness.plan_windows(L)) for _ in range(V))
    print(f'{V} videos x {L} frames of {S}x{S} JPEG = {V * L} frames, {nwin} windows, engine {prec}', flush=True)
    ref = None
    for workers, procs in ((0, False), (8, False), (8, True), (16, True), (0, False), (12, True)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs = harness.run_annotation(eng, anno, tmp, pipe, rng=np.random.RandomState(1), workers=workers, processes=procs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ref = ref or recs
        kind = 'processes' if procs else ('threads' if workers else 'in line')
        print(f'workers={workers:2d} ({kind:9s}): {dt:6.2f} s  {V * L / dt:8.1f} frames/s  {nwin / dt:7.1f} windows/s  identical={recs == ref}', flush=True)
    if os.environ.get('MCG_PROFILE'):   # where the consumer's time goes with the decode off its thread
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        harness.run_annotation(eng, anno, tmp, pipe, rng=np.random.RandomState(1), workers=16, processes=True)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
