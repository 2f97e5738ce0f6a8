"""GPU experiment: the whole forward (mcg_clip_forward: two trunk ranges forked/joined inside + decoder) captured into a HIP graph
and replayed, vs direct launches.  Usage: python tools/lab/graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import synth
from mcgaze_amd.engine import HipEngine
PREC = next((a for a in sys.argv[1:] if a in ('f16x3', 'bf16', 'fp32')), 'f16x3')   # the product engine unless another is named
e = HipEngine(synth.make_state_dict(0), precision=PREC)
img = torch.from_numpy(synth.make_clips(3, 64, 7)).cuda()
out = dict(gaze=torch.empty(4, 448, 3, device='cuda'), boxes=torch.empty(448, 3, 4, device='cuda'), scores=torch.empty(448, 3, device='cuda'))
def direct(iters=20):
    for _ in range(3): e.forward(img, 7, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): e.forward(img, 7, out=out)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
print(f'direct launches: {direct():.3f} ms/step')
ref = {k: v.clone() for k, v in out.items()}
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): e.forward(img, 7, out=out)     # warm-up on the capture stream: side streams probed for it
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        e.forward(img, 7, out=out)
    for v in out.values(): v.zero_()
    g.replay(); torch.cuda.synchronize()
    print('graph replay equals direct:', all(torch.equal(out[k], ref[k]) for k in out))
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); print(f'graph replay: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step')
except Exception as ex:
    print('capture failed:', repr(ex)[:300])
