"""GPU tool: per-launch timing of the contraction kernel over one step (HIP events via
mcg_profile_start/stop).  Usage: python tools/layer_profile.py [clips] [precision]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcgaze_amd import lib as L, synth
from mcgaze_amd.engine import HipEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
lib = L.load()
eng = HipEngine(synth.make_state_dict(0), precision=prec)
img = torch.from_numpy(synth.make_clips(3, B, 7)).cuda()
for _ in range(3):
    eng.forward(img, 7)
torch.cuda.synchronize()
cap = 4096
L.check(lib.mcg_profile_start(cap), 'start')
eng.forward(img, 7)
cnt = C.c_int(); ms = (C.c_float * cap)(); fl = (C.c_double * cap)(); cf = (C.c_int * cap)(); sh = (C.c_int * (3 * cap))()
L.check(lib.mcg_profile_stop(C.byref(cnt), ms, fl, cf, sh, cap), 'stop')
tot = 0.0
print(f'{"#":>3} {"cfg":>3} {"M":>8} {"N":>6} {"K":>6} {"ms":>8} {"TF/s":>7} {"GB/s(min)":>9}')
es = 2 if prec == 'bf16' else 4
for i in range(cnt.value):
    M, N, K = sh[3 * i], sh[3 * i + 1], sh[3 * i + 2]
    tot += ms[i]
    gb = (M * K / (K // min(K, 1e9) if False else 1) * 0 + (M * N + N * K) * es) / 1e9  # lower bound: write out + weights (input reuse varies)
    print(f'{i:3d} {cf[i]:3d} {M:8d} {N:6d} {K:6d} {ms[i]:8.4f} {fl[i] / ms[i] / 1e9:7.1f} {gb / (ms[i] * 1e-3):9.0f}')
print(f'total contraction time {tot:.3f} ms over {cnt.value} launches')
