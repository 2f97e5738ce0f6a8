"""GPU experiment: two concurrent frame ranges of unequal size (do ranges that drift out of lockstep overlap better?  no: even split wins)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['MCG_TRUNK_STREAMS'] = '1'
import torch
from mcgaze_amd import lib as L, synth
from mcgaze_amd.engine import HipEngine, _ptr, _ws
e = HipEngine(synth.make_state_dict(0), precision='bf16')
img = torch.from_numpy(synth.make_clips(3, 64, 7)).cuda()
N, H, W = img.shape[0], 224, 224
def bench(parts, iters=20):
    streams = [torch.cuda.Stream() for _ in parts]
    bufs = []
    for (a, b) in parts:
        n = b - a
        pyr = [torch.empty(n, (H // 4) >> i, (W // 4) >> i, 256, dtype=torch.bfloat16, device='cuda') for i in range(4)]
        bufs.append((n, pyr, (C.c_void_p * 4)(*[p.data_ptr() for p in pyr]), _ws(e.lib.mcg_trunk_workspace_bytes(e._handle, n, H, W, 0), e.device), img[a:b].contiguous()))
    def run():
        for st, (n, pyr, tab, ws, x) in zip(streams, bufs):
            L.check(e.lib.mcg_backbone_fpn_forward(e._handle, C.c_void_p(st.cuda_stream), _ptr(x), n, H, W, 0, tab, _ptr(ws), ws.numel()), 'trunk')
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
for split in (224, 252, 280, 308, 196):
    print(f'split {split}/{N - split}: {bench([(0, split), (split, N)]):.3f} ms')
