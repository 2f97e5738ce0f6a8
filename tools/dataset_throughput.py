"""GPU tool: end-to-end rate of the dataset path (files -> decode -> device preprocessing -> windows -> engine -> merged records,
harness.run_annotation) on a synthetic directory of JPEG frames, with the frames decoded in line and by the look-ahead thread pool.
usage: python tools/dataset_throughput.py [videos=160] [frames_per_video=60] [side=360] [precision=f16x3] [consumers=0]
consumers = K > 0 adds the multi-consumer measurement: K child processes on the SAME GPU, each with its own engine and eight decode helpers, each running
the videos dist.shard_videos gives it (crops seeded per video, so the merged records must equal the single-process run's)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from mcgaze_amd import Config, harness, synth
from mcgaze_amd.engine import HipEngine
from mcgaze_amd.pipeline import DevicePipeline

import json, subprocess
CHILD = len(sys.argv) > 1 and sys.argv[1] == '--child'
if CHILD:
    _, _, c_rank, c_world, c_dir, c_prec = sys.argv[:6]
    sys.argv = sys.argv[:1]
V = int(sys.argv[1]) if len(sys.argv) > 1 else 160
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60
S = int(sys.argv[3]) if len(sys.argv) > 3 else 360
prec = sys.argv[4] if len(sys.argv) > 4 else 'f16x3'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = int(sys.argv[5]) if len(sys.argv) > 5 else 0
vrng = lambda vid: np.random.RandomState(77000 + int(vid))     # per-video crop generator: records independent of the shard layout
if CHILD:
    from mcgaze_amd.dist import shard_videos
    c_rank, c_world = int(c_rank), int(c_world)
    anno = json.load(open(os.path.join(c_dir, 'anno.json')))
    cpipe = DevicePipeline(Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'mcgaze', 'r50_clip7_gaze360.py')).data.test.pipeline)
    ceng = HipEngine(synth.make_state_dict(0), precision=c_prec)
    idx = shard_videos(anno['videos'], c_world, c_rank)
    sub = dict(videos=[anno['videos'][i] for i in idx])
    harness.run_annotation(ceng, dict(videos=sub['videos'][:2]), c_dir, cpipe, workers=4, processes=True, video_rng=vrng)   # warm-up: kernels, allocator, pinned buffers
    torch.cuda.synchronize()
    open(os.path.join(c_dir, f'ready{c_rank}'), 'w').close()
    while not os.path.exists(os.path.join(c_dir, 'go')):
        time.sleep(0.002)
    recs = harness.run_annotation(ceng, sub, c_dir, cpipe, workers=8, processes=True, video_rng=vrng)
    torch.cuda.synchronize()
    json.dump(dict(idx=idx, recs=recs, t_done=time.time()), open(os.path.join(c_dir, f'out{c_rank}.json'), 'w'))
    sys.exit(0)
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
    for workers, procs in (((8, True),) if os.environ.get('MCG_PROFILE') else ((0, False), (8, True)) if K > 0 else ((0, False), (8, False), (8, True), (16, True), (0, False), (12, True))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs = harness.run_annotation(eng, anno, tmp, pipe, rng=np.random.RandomState(1), workers=workers, processes=procs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ref = ref or recs
        kind = 'processes' if procs else ('threads' if workers else 'in line')
        print(f'workers={workers:2d} ({kind:9s}): {dt:6.2f} s  {V * L / dt:8.1f} frames/s  {nwin / dt:7.1f} windows/s  identical={recs == ref}  {harness.last_run_stats}', flush=True)
    if K > 0:
        json.dump(anno, open(os.path.join(tmp, 'anno.json'), 'w'))
        t0 = time.perf_counter()
        single = harness.run_annotation(eng, anno, tmp, pipe, workers=8, processes=True, video_rng=vrng)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t0
        print(f'per-video crop seeds, 1 consumer : {dt1:6.2f} s  {V * L / dt1:8.1f} frames/s', flush=True)
        for k in sorted({2, K}):
            for f in os.listdir(tmp):
                if f.startswith(('ready', 'out')) or f == 'go':
                    os.remove(os.path.join(tmp, f))
            kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--child', str(r), str(k), tmp, prec]) for r in range(k)]
            while not all(os.path.exists(os.path.join(tmp, f'ready{r}')) for r in range(k)):
                if any(c.poll() not in (None, 0) for c in kids):
                    raise SystemExit('a consumer process failed')
                time.sleep(0.01)
            t0 = time.time()
            open(os.path.join(tmp, 'go'), 'w').close()
            for c in kids:
                c.wait()
            outs = [json.load(open(os.path.join(tmp, f'out{r}.json'))) for r in range(k)]
            dt = max(o['t_done'] for o in outs) - t0
            merged = [None] * V
            for o in outs:
                for i, r in zip(o['idx'], o['recs']):
                    merged[i] = r
            same = json.loads(json.dumps(single)) == merged
            print(f'per-video crop seeds, {k} consumers on one GPU: {dt:6.2f} s  {V * L / dt:8.1f} frames/s  records identical to the single process: {same}', flush=True)
    if os.environ.get('MCG_PROFILE'):   # where the consumer's time goes with the decode off its thread
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        harness.run_annotation(eng, anno, tmp, pipe, rng=np.random.RandomState(1), workers=16, processes=True)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
