"""GPU tool: does the fused bottleneck tail run slower right after an MFMA-bound kernel (the package power / clock state the engine's
neighbours leave) than in a loop of its own?  Times mcg_bottleneck_x3 (layer1 <64,1,128>, 448 frames) with HIP events: alone, and
with an FPN-P2-sized f16x3 3x3 convolution (4 ms at the power cap) launched before every call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import engine as E
from mcgaze_amd.packing import bneck_stream
g = torch.Generator().manual_seed(1)
cm, cn, N, H, W = 64, 128, 448, 56, 56
w2 = torch.randn(cm, 3, 3, cm, generator=g) / (9 * cm / 2) ** 0.5
w3 = torch.randn(4 * cm, cm, generator=g) / 8
w1 = torch.randn(cn, 4 * cm, generator=g) / 11
ws, bs = bneck_stream(w2, torch.randn(cm, generator=g) * 0.1, w3, torch.randn(4 * cm, generator=g) * 0.1, w1, torch.randn(cn, generator=g) * 0.1)
ws, bs = ws.cuda(), bs.cuda()
x = torch.randn(N, H, W, cm, device='cuda').relu_()
res = torch.randn(N, H, W, 4 * cm, device='cuda').relu_()

cx = torch.randn(N, H, W, 256, device='cuda')
cw = (torch.randn(256, 3, 3, 256, generator=g) / 48).cuda()
cb = torch.randn(256, generator=g).cuda()


def timed(pre, iters=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        if pre:
            E.conv2d(cx, cw, cb, stride=1, pad=1, relu=False, split=True)
        a.record()
        E.bottleneck_x3(x, res, ws, bs, cn, 1)
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev[5:])
    return t[len(t) // 2], t[0], t[-1]


for _ in range(2):
    for pre in (0, 1):
        med, lo, hi = timed(pre)
        print(f'bneck <64,1,128> {"after an MFMA-bound 3x3 conv" if pre else "in a loop of its own        "}: median {med:.4f} ms (min {lo:.4f}, max {hi:.4f})', flush=True)
