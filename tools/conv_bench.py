"""GPU tool: time one conv shape through mcg_conv2d.
usage: conv_bench.py N H W Cin Cout k stride pad [iters] [residual] [tile] [precision bf16|f16x3|fp32] [data randn|zeros|relu|small]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcgaze_amd import engine as E
N, H, W, Cin, Cout, k, stride, pad = [int(v) for v in sys.argv[1:9]]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 50
res = int(sys.argv[10]) if len(sys.argv) > 10 else 0
tile = int(sys.argv[11]) if len(sys.argv) > 11 else 0
prec = sys.argv[12] if len(sys.argv) > 12 else 'f16x3'
mode = sys.argv[13] if len(sys.argv) > 13 else 'randn'   # the rate depends on the operand bits: the kernel sits on the package power cap
dt = torch.bfloat16 if prec == 'bf16' else torch.float32
split = prec == 'f16x3'
x = torch.randn(N, H, W, Cin, device='cuda')
if mode == 'zeros': x.zero_()
elif mode == 'relu': x.relu_()
elif mode == 'small': x.mul_(1e-3)
x = x.to(dt)
w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(dt)
b = torch.randn(Cout, device='cuda')
Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
r = torch.randn(N, Ho, Wo, Cout, device='cuda').to(dt) if res else None
if split:   # pack once (engine.conv2d packs per call: fine for tests, not for timing), then drive the C-ABI directly
    import ctypes as C
    from mcgaze_amd import lib as L
    from mcgaze_amd.packing import pow2_prescale, split_pack
    lib = L.load()
    ws, wscale = pow2_prescale(w.reshape(Cout, -1).cpu())
    wp = split_pack(ws).cuda()
    y = torch.empty(N, Ho, Wo, Cout, device='cuda')
    d = L.ConvDesc(x.data_ptr(), wp.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), N, H, W, Cin, Cout, k, k, stride, pad, 1,
                   1 if res else 0, 0, 0, None, 0, 1, 0, 0, tile, 0, wscale)
    s_ = E._stream()
    run = lambda: L.check(lib.mcg_conv2d(s_, L.MCG_F16X3, C.byref(d)), 'mcg_conv2d')
else:
    run = lambda: E.conv2d(x, w, b, stride=stride, pad=pad, relu=True, residual=r, residual_mode=1 if res else 0, tile=tile)
for _ in range(60):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
fl = 2.0 * N * Ho * Wo * Cout * Cin * k * k
print(f'conv N={N} {H}x{W} {Cin}->{Cout} k{k} s{stride}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TF/s  (tile={tile or "auto"}, {prec}, data={mode})')
