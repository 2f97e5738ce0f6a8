"""GPU tool: trunk (backbone + FPN) time for 64 clips with an engine option off / on.
usage: python tools/lab/trunk_time.py [option=pointwise_pair] [precision=bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import synth
from mcgaze_amd.engine import HipEngine

opt = sys.argv[1] if len(sys.argv) > 1 else 'pointwise_pair'
prec = sys.argv[2] if len(sys.argv) > 2 else 'f16x3'   # the product engine; 'bf16' = the throughput mode
eng = HipEngine(synth.make_state_dict(0), precision=prec)
img = torch.from_numpy(synth.make_clips(3, 64, 7)).cuda()
for streams in (1, 2):
    eng.set_option('trunk_streams', streams)
    for val in (0, 1, 0, 1):
        eng.set_option(opt, val)
        for _ in range(3):
            eng.backbone_fpn(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.backbone_fpn(img)
        torch.cuda.synchronize()
        print(f'trunk_streams={streams} {opt}={val}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms', flush=True)
