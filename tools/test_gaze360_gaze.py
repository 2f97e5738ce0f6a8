#!/usr/bin/env python
"""Dataset inference with the reference's command line (its tools/test_gaze360_gaze.py:20-44):

    python tools/test_gaze360_gaze.py <config> <checkpoint> --json data/gaze360/test.json --root data/gaze360/test_rawframes/ \
        [--device cuda:0] [--cfg-options k=v ...] [--precision f16x3|fp32|f16|bf16] [--batch-clips 64] [--seed S] [--anno gt.json]

Every video of the annotation file is cut into 7-frame windows (stride 4), the frames of each window go through
cfg.data.test.pipeline on the device (mcgaze_amd.pipeline), all windows run through the HIP engine in batches of --batch-clips
clips, overlaps are merged per video and results/results_<config>_<json> is written with the reference's schema
(mcgaze_amd.harness).  With --anno the MAE of tools/calculate_mae_gaze360.py / calculate_mae_l2cs.py is printed too
(mcgaze_amd.metric).  Under torch.distributed.run the videos are sharded over the ranks by frame count and rank 0 writes the file.
--seed fixes the crop draws of CenterCrop(crop_type='relative_range'), which the reference leaves to the unseeded global RNG
(SURVEY.md section 5)."""
import argparse
import zlib
import json
import os
import sys
import time

# multi-process GPU work on this ROCm stack needs dmabuf IPC (RCCL / tensor sharing across ranks fail with
# `hipIpcGetMemHandle: invalid argument` otherwise); set before the runtime loads, never overriding the caller
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcgaze_amd import harness, init_detector, metric  # noqa: E402
from mcgaze_amd.dist import gather_records, shard_videos  # noqa: E402
from mcgaze_amd.pipeline import DevicePipeline  # noqa: E402


def parse_value(v):
    for cast in (int, float):
        try:
            return cast(v)
        except ValueError:
            pass
    if v.lower() in ('true', 'false'):
        return v.lower() == 'true'
    if v.startswith('[') or v.startswith('('):
        return eval(v, {}, {})
    return [parse_value(x) for x in v.split(',')] if ',' in v else v


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('config', help='Config file')
    ap.add_argument('checkpoint', help='Checkpoint file')
    ap.add_argument('--json', default='data/gaze360/test.json', help='Path to gaze test json file')
    ap.add_argument('--root', default='data/gaze360/test_rawframes/', help='Path to image file')
    ap.add_argument('--device', default='cuda:0', help='Device used for inference')
    ap.add_argument('--cfg-options', nargs='+', default=None, help='k=v overrides merged into the config (mmcv DictAction syntax)')
    ap.add_argument('--precision', default='f16x3', choices=['f16x3', 'fp32', 'f16', 'bf16'],
                    help="'f16x3' (default): parity-grade fast engine; 'fp32': exact reference mode; 'f16' / 'bf16': 16-bit throughput modes in fp16 / bf16 (not within the 1e-3 tolerance; fp16 is eight times closer)")
    ap.add_argument('--batch-clips', type=int, default=64)
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--per-video-seed', action='store_true',
                    help='seed the crop draws per VIDEO (seed * 1000003 + video id; a non-integer id enters as crc32 of its text) instead of one generator per process: the records then do not depend on the number of ranks')
    ap.add_argument('--workers', type=int, default=8, help='decode helpers that run ahead of the GPU (0 = decode in line); 8 processes measured 9 500 - 9 800 frames/s against 1 900 - 2 000 in line (DESIGN.md section 4)')
    ap.add_argument('--decode', default='processes', choices=['processes', 'threads'], help="kind of decode helper: child processes writing into a /dev/shm ring (default), or host threads")
    ap.add_argument('--ranks-per-gpu', type=int, default=1, help='with torch.distributed.run: consecutive ranks that share one GPU (one consumer process now keeps the device ~80 % busy: more than one per GPU measured slower on the MI355X box; kept for hosts with slower cores)')
    ap.add_argument('--anno', default=None, help='ground-truth annotation json: print the MAE')
    ap.add_argument('--setting', default=None, choices=['gaze360', 'l2cs'], help='metric variant (default: from the config name)')
    a = ap.parse_args(argv)
    if a.cfg_options is not None:
        a.cfg_options = {kv.split('=', 1)[0]: parse_value(kv.split('=', 1)[1]) for kv in a.cfg_options}
    return a


def _video_key(vid):
    """--per-video-seed: what a video id adds to the seed.  An INTEGER id (Gaze360's) enters as itself -- round 4's records reproduce --;
    any other id (a string name) through crc32 of its text (round 5 had switched every id to crc32, silently changing the draws of integer
    ids: ADVICE r5)."""
    if isinstance(vid, (int, np.integer)) or (isinstance(vid, str) and vid.lstrip('-').isdigit()):
        return int(vid)
    return zlib.crc32(str(vid).encode())


def main(argv=None):
    a = parse_args(argv)
    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    device = a.device if world == 1 else f'cuda:{local // max(a.ranks_per_gpu, 1)}'
    d = torch.device(device)
    torch.cuda.set_device(d.index if d.index is not None else 0)   # 'cuda' without an index is device 0 (set_device rejects it)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(torch.device(device))
        # several ranks on one GPU cannot form an RCCL communicator (one rank per device): the record gather is a host-side object
        # gather anyway, so gloo carries it then
        if a.ranks_per_gpu > 1:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(device))
    print(time.strftime('%Y-%m-%d %H:%M:%S', time.localtime(time.time())))
    t_start = time.time()
    model = init_detector(a.config, a.checkpoint, device=device, cfg_options=a.cfg_options, precision=a.precision)
    pipe = DevicePipeline(model.cfg.data.test.pipeline)
    anno = json.load(open(a.json))
    idx = shard_videos(anno['videos'], world, rank)
    t_init = time.time()
    rng = np.random.RandomState(a.seed + rank) if a.seed is not None else None
    recs = harness.run_annotation(model.engine(), dict(videos=[anno['videos'][i] for i in idx]), a.root, pipe, batch_clips=a.batch_clips, rng=rng, workers=a.workers,
                                  processes=a.decode == 'processes',
                                  video_rng=(lambda vid: np.random.RandomState(((a.seed or 0) * 1000003 + _video_key(vid)) & 0x7fffffff)) if a.per_video_seed else None)
    torch.cuda.synchronize()
    t_run = time.time()
    if world > 1:
        recs = gather_records(idx, recs, len(anno['videos']))
    if rank == 0:
        path = harness.dump_results(recs, a.config, a.json)
        print('Done', path)
        if a.anno:
            setting = a.setting or ('l2cs' if 'l2cs' in os.path.basename(a.config) else 'gaze360')
            metric.gaze_error(recs, json.load(open(a.anno)), 'fusion_gazes', setting=setting)
    if rank == 0:
        frames = sum(len(anno['videos'][i]['file_names']) for i in idx)
        print(f'[mcgaze_amd] model + annotation {t_init - t_start:.2f} s, inference {t_run - t_init:.2f} s ({frames / max(t_run - t_init, 1e-9):.0f} frames/s on this rank), '
              f'gather + result file + metric {time.time() - t_run:.2f} s', file=sys.stderr)
    print(time.strftime('%Y-%m-%d %H:%M:%S', time.localtime(time.time())))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
