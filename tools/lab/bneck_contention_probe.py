"""GPU tool: does the fused bottleneck tail (mcg_bottleneck_x3) reproduce its own result bit for bit while ANOTHER thread keeps the GPU busy
(a second engine's forwards on its own stream)?  Reports where y / z differ.  usage: bneck_contention_probe.py [cm=128] [cn=128] [iters=150] [busy=1]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import engine as E, synth
from mcgaze_amd.packing import bneck_stream
a = [int(v) for v in sys.argv[1:]]
cm, cn, iters, busy = (a + [128, 128, 3000, 1][len(a):])[:4]
N, H, W = 70, (28 if cm == 128 else 56), (28 if cm == 128 else 56)
g = torch.Generator().manual_seed(1)
w2 = torch.randn(cm, 3, 3, cm, generator=g) / (9 * cm / 2) ** 0.5
w3 = torch.randn(4 * cm, cm, generator=g) / 8
w1 = torch.randn(cn, 4 * cm, generator=g) / 11 if cn else None
ws, bs = bneck_stream(w2, torch.randn(cm, generator=g) * 0.1, w3, torch.randn(4 * cm, generator=g) * 0.1, w1, torch.randn(cn, generator=g) * 0.1 if cn else None)
ws, bs = ws.cuda(), bs.cuda()
x = torch.randn(N, H, W, cm, device='cuda').relu_()
res = torch.randn(N, H, W, 4 * cm, device='cuda').relu_()
ref = [t.clone() for t in E.bottleneck_x3(x, res, ws, bs, cn, 1) if t is not None]
torch.cuda.synchronize()
stop = False
started = threading.Event()


def noise():
    eng = E.HipEngine(synth.make_state_dict(0), precision='f16x3')
    img = torch.from_numpy(synth.make_clips(7, 10, 7)).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            eng.forward(img, 7)
            s.synchronize()
            started.set()


th = threading.Thread(target=noise)
if busy:
    th.start()
    started.wait()
s = torch.cuda.Stream()
bad = 0
with torch.cuda.stream(s):
    for it in range(iters):
        out = [t for t in E.bottleneck_x3(x, res, ws, bs, cn, 1) if t is not None]
        s.synchronize()
        for name, o, r in zip('yz', out, ref):
            if not torch.equal(o, r):
                d = (o - r).abs()
                idx = torch.nonzero(d.amax(dim=3) > 0)
                ch = torch.nonzero(d.amax(dim=(0, 1, 2)) > 0).flatten()
                bad += 1
                if bad <= 12:
                    px = sorted(set((int(f), int(yy), int(xx)) for f, yy, xx in idx.tolist()))
                    print(f'iter {it} {name}: {idx.shape[0]} pixels, max {float(d.max()):.2e} (scale {float(r.abs().max()):.1f}), channels {int(ch.min())}..{int(ch.max())} ({ch.numel()}), (frame, row, col): {px[:14]}', flush=True)
stop = True
if busy:
    th.join()
print(f'cm={cm} cn={cn} busy={busy}: {bad} mismatching outputs in {iters} launches')
