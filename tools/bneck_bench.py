"""GPU tool: time the fused bottleneck tail (mcg_bottleneck_x3) alone; optional phase trace of workgroup 0.
usage: bneck_bench.py [frames=448] [H=56] [W=56] [nsrc=1] [cn=64] [iters=50] [cm=64] [trace=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcgaze_amd import engine as E
from mcgaze_amd.packing import bneck_stream
a = [int(v) for v in sys.argv[1:]]
N, H, W, nsrc, cn, iters, cm, trace = (a + [448, 56, 56, 1, 64, 50, 64, 0][len(a):])[:8]
g = torch.Generator().manual_seed(1)
w2 = torch.randn(cm, 3, 3, cm, generator=g) / (9 * cm / 2) ** 0.5
w3 = torch.randn(4 * cm, cm + 64 * (nsrc - 1), generator=g) / 8
w1 = torch.randn(cn, 4 * cm, generator=g) / 11 if cn else None
ws, bs = bneck_stream(w2, torch.randn(cm, generator=g) * 0.1, w3, torch.randn(4 * cm, generator=g) * 0.1, w1, torch.randn(cn, generator=g) * 0.1 if cn else None)
ws, bs = ws.cuda(), bs.cuda()
x = torch.randn(N, H, W, cm, device='cuda').relu_()
src2 = torch.randn(N, H, W, 4 * cm if nsrc == 1 else 64, device='cuda').relu_()
for _ in range(10):
    E.bottleneck_x3(x, src2, ws, bs, cn, nsrc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    E.bottleneck_x3(x, src2, ws, bs, cn, nsrc)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
M = N * H * W
gb = 4.0 * M * (cm + (4 * cm if nsrc == 1 else 64) + 4 * cm + cn) / 1e9
fl = 2.0 * M * (9 * cm * cm + (cm + 64 * (nsrc - 1)) * 4 * cm + 4 * cm * cn)
print(f'bneck_x3 {N}x{H}x{W} cm={cm} nsrc={nsrc} cn={cn}: {ms:.4f} ms  {gb / ms:.2f} TB/s algorithmic  {fl / ms / 1e9:.1f} TF/s algorithmic')
if trace:
    tr = torch.zeros(4096, dtype=torch.int64, device='cuda')
    E.bottleneck_x3(x, src2, ws, bs, cn, nsrc, trace=tr)
    torch.cuda.synchronize()
    t = tr.cpu().tolist()
    nch = 4 * cm // 64
    per = 2 + nch                                  # stamps per tile of the compute wave: start, after conv2, after each chunk
    c = [v for v in t[:2048] if v]
    print('compute wave 0, first tiles (cycles): conv2 phase | each chunk ... | tile total')
    for i in range(min(4, len(c) // per)):
        s = c[i * per:(i + 1) * per]
        nxt = c[(i + 1) * per] if (i + 1) * per < len(c) else s[-1]
        print(f'  tile {i}: conv2 {s[1] - s[0]:7d} | chunks ' + ' '.join(f'{s[k + 1] - s[k]:6d}' for k in range(1, per - 1)) + f' | total {nxt - s[0]:7d}')
    l = [v for v in t[2048:] if v]
    ns = 9 * (cm // 64) ** 2 + nch * (cm // 64 + nsrc - 1 + cn // 64)
    print(f'loader wave, slabs per tile = {ns}: (wait-for-landing done -> barrier passed) per slab, second tile')
    rows = []
    for k in range(ns, min(2 * ns, len(l) // 2 - 1)):
        rows.append((l[2 * k] - l[2 * k - 1], l[2 * k + 1] - l[2 * k]))          # (issue + landing wait, barrier wait)
    print('  issue+landing wait: ' + ' '.join(f'{a_:5d}' for a_, _ in rows))
    print('  barrier wait:       ' + ' '.join(f'{b_:5d}' for _, b_ in rows))
