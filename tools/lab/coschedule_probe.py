"""GPU probe: do an MFMA-bound conv and an HBM-bound conv overlap when they run on two streams?  (Every kernel is sized to fill a CU,
so two frame ranges overlap only at kernel tails; this measures what a half-CU footprint would buy.)
usage: python tools/lab/coschedule_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import engine as E

dev = 'cuda'
def mk(N, H, W, Cin, Cout, k, res):
    x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    r = torch.randn(N, H, W, Cout, device=dev).to(torch.bfloat16) if res else None
    return x, w, b, r
A = mk(224, 56, 56, 256, 256, 3, False)     # MFMA-bound: FPN P2 3x3, one frame range
B = mk(224, 56, 56, 64, 256, 1, True)       # HBM-bound: layer1 conv3 + residual
def run(op, tile, n):
    x, w, b, r = op
    for _ in range(n):
        E.conv2d(x, w, b, pad=(w.shape[1] // 2), relu=True, residual=r, residual_mode=1 if r is not None else 0, tile=tile)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3
n = 20
for ta in (14, 9):
    for tb in (9,):
        a = timed(lambda: run(A, ta, n)) / n
        b = timed(lambda: run(B, tb, 3 * n)) / (3 * n)
        def both():
            with torch.cuda.stream(s1): run(A, ta, n)
            with torch.cuda.stream(s2): run(B, tb, 3 * n)
        c = timed(both)
        print(f'A tile {ta}: {a:.3f} ms, B tile {tb}: {b:.3f} ms; {n} x A alone + {3 * n} x B alone = {n * a + 3 * n * b:.2f} ms, concurrently {c:.2f} ms')
