"""GPU experiment: which pairs of HIP streams give the two-range trunk its overlap?  Streams of one priority are packed onto
GPU_MAX_HW_QUEUES (4) hardware queues; this times the trunk with range 0 on stream i and range 1 on stream j for a list of
streams created in order.  Usage: python tools/stream_pairs.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['MCG_TRUNK_STREAMS'] = '1'
import torch
from mcgaze_amd import lib as L, synth
from mcgaze_amd.engine import HipEngine, _ptr, _ws
e = HipEngine(synth.make_state_dict(0), precision='bf16')
img = torch.from_numpy(synth.make_clips(3, 64, 7)).cuda()
N, H, W = img.shape[0], 224, 224
n = N // 2
streams = [torch.cuda.Stream() for _ in range(9)] + [torch.cuda.Stream(priority=-1) for _ in range(2)]
bufs = []
for i in range(2):
    pyr = [torch.empty(n, (H // 4) >> j, (W // 4) >> j, 256, dtype=torch.bfloat16, device='cuda') for j in range(4)]
    bufs.append(((C.c_void_p * 4)(*[p.data_ptr() for p in pyr]), _ws(e.lib.mcg_trunk_workspace_bytes(e._handle, n, H, W, 0), e.device), img[i * n:(i + 1) * n].contiguous(), pyr))
def run(si, sj, iters=10):
    def once():
        for k, st in enumerate((streams[si], streams[sj])):
            tab, ws, x, _ = bufs[k]
            L.check(e.lib.mcg_backbone_fpn_forward(e._handle, C.c_void_p(st.cuda_stream), _ptr(x), n, H, W, 0, tab, _ptr(ws), ws.numel()), 'trunk')
    for _ in range(2): once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): once()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
print('same stream (0,0): %.2f ms' % run(0, 0))
for j in range(1, 11):
    print(f'streams (0,{j}){" [j high priority]" if j >= 9 else ""}: {run(0, j):.2f} ms')
print('streams (9,10) both high priority: %.2f ms' % run(9, 10))
print('streams (1,2): %.2f  (1,5): %.2f  (2,6): %.2f  (3,4): %.2f' % (run(1, 2), run(1, 5), run(2, 6), run(3, 4)))
