"""GPU experiment: two independent 224-frame trunk pipelines on two streams, in lockstep vs staggered by a fraction of a trunk
(does a memory-bound layer1 of one pipeline overlap better with the MFMA-bound FPN / layer3-4 convs of the other?).
Usage: MCG_TRUNK_STREAMS=1 python tools/trunk_phase.py"""
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
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bufs = []
for i in range(2):
    pyr = [torch.empty(n, (H // 4) >> j, (W // 4) >> j, 256, dtype=torch.bfloat16, device='cuda') for j in range(4)]
    bufs.append((pyr, (C.c_void_p * 4)(*[p.data_ptr() for p in pyr]), _ws(e.lib.mcg_trunk_workspace_bytes(e._handle, n, H, W, 0), e.device), img[i * n:(i + 1) * n].contiguous()))
def trunk(i):
    pyr, tab, ws, x = bufs[i]
    L.check(e.lib.mcg_backbone_fpn_forward(e._handle, C.c_void_p(streams[i].cuda_stream), _ptr(x), n, H, W, 0, tab, _ptr(ws), ws.numel()), 'trunk')
def run(iters, stagger_frames):
    # stagger: stream 1 first runs a partial-size dummy trunk so that it lags stream 0 by that fraction
    torch.cuda.synchronize()
    if stagger_frames:
        pyr, tab, ws, x = bufs[1]
        L.check(e.lib.mcg_backbone_fpn_forward(e._handle, C.c_void_p(streams[1].cuda_stream), _ptr(x), stagger_frames, H, W, 0, tab, _ptr(ws), ws.numel()), 'trunk')
    t0 = time.perf_counter()
    for _ in range(iters):
        trunk(0); trunk(1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
for _ in range(3): run(3, 0)
for st in (0, 56, 112, 168, 0, 112):
    print(f'stagger {st:3d} frames: {run(30, st):.3f} ms per 448 frames')
