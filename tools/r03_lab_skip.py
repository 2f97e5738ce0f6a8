"""LAB (GPU): what a layer class costs the PIPELINED step -- bench.Leg timed with classes of launches skipped (wrong results, timing only).
usage: python tools/r03_lab_skip.py [precision] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mcgaze_amd import synth
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
argv, sys.argv = sys.argv, ['bench.py']
a = bench.parse()
sys.argv = argv
dev = torch.device('cuda', 0)
B, T = 64, 7
img = torch.from_numpy(synth.make_clips(3, B, T)).to(dev)
leg = bench.Leg(a, prec, dev, 1, 0, None, img, B, T)
L1, L2, L3, L4 = 8, 16, 32, 64
cases = [('none', 0), ('none', 0), ('layer1 conv1', L1 | 1), ('layer1 conv2', L1 | 2), ('layer1 conv3', L1 | 4), ('layer1 all', L1 | 7),
         ('layer2 conv1', L2 | 1), ('layer2 conv2', L2 | 2), ('layer2 conv3', L2 | 4), ('layer2 all', L2 | 7),
         ('layer3 conv3', L3 | 4), ('layer3 all', L3 | 7), ('layer4 all', L4 | 7), ('laterals', 128), ('fpn 3x3', 256), ('none', 0)]
for streams in (2, 1):
    a.trunk_streams = streams
    leg.eng.set_option('trunk_streams', streams)
    base = None
    for name, mask in cases:
        leg.eng.set_option('lab_skip', mask)
        el = leg.timed(steps, 3)
        ms = el / steps * 1e3
        if base is None and name == 'none':
            pass
        if name == 'none':
            base = ms
        print(f'trunk_streams={streams} skip {name:14s}: {ms:7.3f} ms/step  (delta {ms - base:+.3f})', flush=True)
leg.eng.set_option('lab_skip', 0)
