"""GPU tool: does the PLACEMENT of the fused tail's four streams (x, residual, y, z) in memory matter?  One arena, slices at the
offsets the engine's trunk workspace gives them (every buffer a multiple of 2 MiB from the others: trunk_layout, engine.hip) against
the same slices skewed by odd multiples of a few KiB.  Layer1 <64,1,128> at 448 frames, HIP events, median of 40."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import lib as L
from mcgaze_amd.engine import _ptr, _stream
from mcgaze_amd.packing import bneck_stream
lib = L.load()
g = torch.Generator().manual_seed(1)
cm, cn, N, H, W = 64, 128, 448, 56, 56
w2 = torch.randn(cm, 3, 3, cm, generator=g) / (9 * cm / 2) ** 0.5
w3 = torch.randn(4 * cm, cm, generator=g) / 8
w1 = torch.randn(cn, 4 * cm, generator=g) / 11
ws, bs = bneck_stream(w2, torch.randn(cm, generator=g) * 0.1, w3, torch.randn(4 * cm, generator=g) * 0.1, w1, torch.randn(cn, generator=g) * 0.1)
ws, bs = ws.cuda(), bs.cuda()
M = N * H * W
big = M * 1024                     # bytes of a 256-channel f32 map
arena = torch.empty(6 * big + (64 << 20), dtype=torch.uint8, device='cuda')
base = (-arena.data_ptr()) % (1 << 21)          # 2 MiB-align the arena's origin like a fresh hipMalloc


def view(off, ch):
    return arena[base + off: base + off + M * ch * 4].view(torch.float32).view(N, H, W, ch)


def run(name, offs):
    x, res, y, z = view(offs[0], cm), view(offs[1], 4 * cm), view(offs[2], 4 * cm), view(offs[3], cn)
    x.normal_().relu_(); res.normal_().relu_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(45)]
    for a, b in ev:
        a.record()
        L.check(lib.mcg_bottleneck_x3(_stream(), _ptr(x), _ptr(res), _ptr(ws), _ptr(bs), _ptr(y), _ptr(z), N, H, W, cm, 1, cn, None), 'bneck')
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev[5:])
    print(f'{name}: median {t[len(t) // 2]:.4f} ms (min {t[0]:.4f})  offsets mod 2 MiB: {[o % (1 << 21) for o in offs]}', flush=True)


# the engine's relative placement: xa | xb | o1 | o2 | ... | c0  (x = o1, res = xb, y = c0, z = o2)
eng = [2 * big, 1 * big, 4 * big + big // 2, 2 * big + big // 2]
for rep in range(2):
    run('engine-like (all 2 MiB-congruent)', eng)
    run('skewed by 4 KiB x {0, 1, 2, 3}   ', [eng[0], eng[1] + 4096, eng[2] + 8192, eng[3] + 12288])
    run('skewed by 68 KiB x {0, 1, 2, 3}  ', [eng[0], eng[1] + 69632, eng[2] + 2 * 69632, eng[3] + 3 * 69632])
    run('skewed by 1 MiB + 260 KiB steps   ', [eng[0], eng[1] + 1314816, eng[2] + 2 * 1314816 % (1 << 21), eng[3] + 3 * 1314816 % (1 << 21)])
