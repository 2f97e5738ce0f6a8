"""GPU tool: trunk (backbone + FPN) time for 64 clips under combinations of the engine options trunk_streams x max_range_frames
(MALL-sized frame chunks vs whole-batch ranges).  usage: python tools/lab/trunk_chunk_sweep.py [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import synth
from mcgaze_amd.engine import HipEngine

prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'   # the product engine; 'bf16' = the throughput mode
eng = HipEngine(synth.make_state_dict(0), precision=prec)
img = torch.from_numpy(synth.make_clips(3, 64, 7)).cuda()
for streams in (1, 2, 3, 4):
    for cap in (0, 112, 56, 28):
        eng.set_option('trunk_streams', streams); eng.set_option('max_range_frames', cap)
        for _ in range(3):
            eng.backbone_fpn(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.backbone_fpn(img)
        torch.cuda.synchronize()
        print(f'streams={streams} max_range_frames={cap or "none"}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms', flush=True)
