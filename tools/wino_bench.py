"""GPU tool: wino_x3.hpp (1-D Winograd F(2,3), f16x3) against the direct f16x3 contraction kernel on one 3x3 / stride 1 conv shape.
usage: wino_bench.py N H W Cin Cout [iters] [data randn|relu] [tile 0..3] [g 2|4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from mcgaze_amd import engine as E, lib as L
from mcgaze_amd.packing import wino_pack, split_pack
N, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 30
mode = sys.argv[7] if len(sys.argv) > 7 else 'randn'
TILE = int(sys.argv[8]) if len(sys.argv) > 8 else 0
GG = int(sys.argv[9]) if len(sys.argv) > 9 else 2   # 2 = F(2,3), 4 = F(4,3)
lib = L.load()
x = torch.randn(N, H, W, Cin, device='cuda')
if mode == 'relu': x.relu_()
w = torch.randn(Cout, 3, 3, Cin) / (9 * Cin) ** 0.5
b = torch.randn(Cout, device='cuda')
u = wino_pack(w, g=GG).cuda()
ws = split_pack(w.reshape(Cout, -1)).cuda()
y = torch.empty(N, H, W, Cout, device='cuda')
yd = torch.empty_like(y)
s = E._stream()
def wino():
    L.check(lib.mcg_conv3x3_wino_x3(s, E._ptr(x), E._ptr(u), E._ptr(b), E._ptr(y), N, H, W, Cin, Cout, 1, TILE, 0.0, GG), 'wino')
d = L.ConvDesc(x.data_ptr(), ws.data_ptr(), b.data_ptr(), None, yd.data_ptr(), N, H, W, Cin, Cout, 3, 3, 1, 1, 1, 0, 0, 0, None, 0, 1, 0, 0, 0, 0)
def direct():
    L.check(lib.mcg_conv2d(s, L.MCG_F16X3, C.byref(d)), 'direct')
fl = 2.0 * N * H * W * Cout * Cin * 9
for name, fn in (('direct', direct), ('wino', wino), ('direct', direct), ('wino', wino)):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f'{name:7s} N={N} {H}x{W} {Cin}->{Cout}: {ms:.4f} ms  {fl / ms / 1e9:.1f} algorithmic TF/s ({mode})', flush=True)
print('max |wino - direct| / max|direct| =', float((y - yd).abs().max() / yd.abs().max()))
