"""GPU tool: trunk-only (backbone + FPN) step time for several frame-chunk sizes.  Usage: python tools/trunk_time.py [chunks...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcgaze_amd import synth
from mcgaze_amd.engine import HipEngine
e = HipEngine(synth.make_state_dict(0), precision='bf16')
img = torch.from_numpy(synth.make_clips(3, 64, 7)).cuda()
for ch in [int(a) for a in sys.argv[1:]] or [0]:
    for _ in range(5): e.backbone_fpn(img, ch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): e.backbone_fpn(img, ch)
    torch.cuda.synchronize(); print(f'trunk only, chunk_frames={ch}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step')
