"""GPU tool: per-launch timing of the contraction kernel over one step (HIP events via
mcg_engine_profile_start/stop), trunk on one stream.  Usage: python tools/layer_profile.py [clips] [precision] [tile]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcgaze_amd import lib as L, synth
from mcgaze_amd.engine import HipEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else 'f16x3'   # the product engine; 'bf16' = the throughput mode
eng = HipEngine(synth.make_state_dict(0), precision=prec)
eng.set_option('trunk_streams', 1)
if len(sys.argv) > 3:
    eng.set_option('tile', int(sys.argv[3]))
img = torch.from_numpy(synth.make_clips(3, B, 7)).cuda()
for _ in range(3):
    eng.forward(img, 7)
torch.cuda.synchronize()
eng.profile_start(4096)
eng.forward(img, 7)
torch.cuda.synchronize()
rec = eng.profile_stop(4096)
tot = 0.0
print(f'{"#":>3} {"cfg":>3} {"M":>8} {"N":>6} {"K":>6} {"ms":>8} {"TF/s":>7} {"algoGB":>7} {"GB/s":>6}')
for i, (ms, fl, cf, (M, N, K), by) in enumerate(rec):
    tot += ms
    print(f'{i:3d} {cf:3d} {M:8d} {N:6d} {K:6d} {ms:8.4f} {fl / ms / 1e9:7.1f} {by / 1e9:7.3f} {by / 1e9 / (ms * 1e-3):6.0f}')
print(f'total contraction time {tot:.3f} ms over {len(rec)} launches')
