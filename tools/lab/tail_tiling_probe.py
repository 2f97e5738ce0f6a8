"""GPU probe: tail-aware tiling.  A layer whose 256x256 tile count is 1.34 x the CU count (layer3 at 448 frames: 343 tiles) runs two
rounds, the second one a third full.  Split the rows into a head that fills whole rounds with big tiles and a tail run with small
tiles (all tiles walk K in the same order -> same bits).  usage: python tools/lab/tail_tiling_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import engine as E

def bench(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

for (H, Cin, Cout, k) in ((14, 256, 256, 3), (14, 1024, 256, 1), (14, 256, 1024, 1), (7, 512, 512, 3)):
    N = 448
    x = torch.randn(N, H, H, Cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device='cuda')
    full = bench(lambda: E.conv2d(x, w, b, pad=k // 2, relu=True))
    rows = N * H * H
    tiles = (rows + 255) // 256 * ((Cout + 255) // 256)
    best = None
    for head_tiles in (256, 512):
        nh = min(N, int(head_tiles / ((Cout + 255) // 256) * 256 / (H * H)))
        if nh <= 0 or nh >= N: continue
        xh, xt = x[:nh].contiguous(), x[nh:].contiguous()
        for tt in (9, 15, 11):
            t = bench(lambda: (E.conv2d(xh, w, b, pad=k // 2, relu=True), E.conv2d(xt, w, b, pad=k // 2, relu=True, tile=tt)))
            print(f'  {H}x{H} {Cin}->{Cout} k{k}: head {nh} frames (auto tile) + tail {N - nh} frames tile {tt}: {t:.4f} ms')
    print(f'{H}x{H} {Cin}->{Cout} k{k}: {tiles} tiles of 256x256, one launch: {full:.4f} ms')
