"""GPU tool: stem (conv7x7/s2 + BN + ReLU + max-pool) time, fused kernel vs the three-kernel path.  Usage: python tools/stem_bench.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcgaze_amd import engine as E, synth
from mcgaze_amd.packing import PackedWeights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 448
pw = PackedWeights(synth.make_state_dict(0), dtype=torch.bfloat16)
img = torch.from_numpy(synth.make_clips(3, n // 7, 7)).cuda()
for fused in ('0', '1'):
    os.environ['MCG_STEM_FUSED'] = fused
    for _ in range(10): E.stem(img, pw.stem['w'], pw.stem['bias'], torch.bfloat16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): E.stem(img, pw.stem['w'], pw.stem['bias'], torch.bfloat16)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    gb = (img.numel() * 4 + n * 56 * 56 * 64 * 2) / 1e9
    print(f'stem fused={fused}: {dt * 1e3:.3f} ms  ({gb / dt:.0f} GB/s of the algorithmic {gb:.2f} GB, {2 * n * 112 * 112 * 64 * 147 / dt / 1e12:.0f} TFLOP/s)')
