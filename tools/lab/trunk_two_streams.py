"""GPU experiment: the trunk over 448 frames as one launch sequence vs two half-batches on two concurrent HIP streams
(do the tails of one stream's kernels fill with the other's blocks?).  Usage: python tools/lab/trunk_two_streams.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import lib as L, synth
from mcgaze_amd.engine import HipEngine, _ptr, _ws
PREC = next((a for a in sys.argv[1:] if a in ('f16x3', 'bf16', 'fp32')), 'f16x3')   # the product engine unless another is named
e = HipEngine(synth.make_state_dict(0), precision=PREC)
img = torch.from_numpy(synth.make_clips(3, 64, 7)).cuda()
N, H, W = img.shape[0], 224, 224
def bench(parts, iters=20):
    streams = [torch.cuda.Stream() for _ in parts]
    bufs = []
    for (a, b) in parts:
        n = b - a
        pyr = [torch.empty(n, (H // 4) >> i, (W // 4) >> i, 256, dtype=torch.bfloat16, device='cuda') for i in range(4)]
        tab = (C.c_void_p * 4)(*[p.data_ptr() for p in pyr])
        ws = _ws(e.lib.mcg_trunk_workspace_bytes(e._handle, n, H, W, 0), e.device)
        bufs.append((n, pyr, tab, ws, img[a:b].contiguous()))
    def run():
        for st, (n, pyr, tab, ws, x) in zip(streams, bufs):
            L.check(e.lib.mcg_backbone_fpn_forward(e._handle, C.c_void_p(st.cuda_stream), _ptr(x), n, H, W, 0, tab, _ptr(ws), ws.numel()), 'trunk')
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
print(f'one stream, 448 frames: {bench([(0, N)]):.3f} ms')
print(f'two streams, 224 + 224: {bench([(0, N // 2), (N // 2, N)]):.3f} ms')
print(f'four streams, 112 x 4:  {bench([(i * N // 4, (i + 1) * N // 4) for i in range(4)]):.3f} ms')
