"""GPU tool: decoder-only (4 x [RoIAlign + stage] + gaze head) step time on precomputed pyramids.  Usage: python tools/decoder_time.py [iters] [precision=f16x3] [clips=64] [engine option NAME=INT ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcgaze_amd import lib as L, synth
from mcgaze_amd.engine import HipEngine, _ptr, _ws, _stream
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prec = sys.argv[2] if len(sys.argv) > 2 else 'f16x3'   # the product engine; 'bf16' = the throughput mode
clips = int(sys.argv[3]) if len(sys.argv) > 3 else 64
e = HipEngine(synth.make_state_dict(0), precision=prec)
for kv in sys.argv[4:]:
    e.set_option(kv.split('=')[0], int(kv.split('=')[1]))
img = torch.from_numpy(synth.make_clips(3, clips, 7)).cuda()
N, T, H, W = img.shape[0], 7, 224, 224
pyr = e.backbone_fpn(img)
tab = (C.c_void_p * 4)(*[p.data_ptr() for p in pyr])
ws = _ws(e.lib.mcg_decoder_workspace_bytes(e._handle, N), e.device)
out = dict(gaze=torch.empty(4, N, 3, device='cuda'), boxes=torch.empty(N, 3, 4, device='cuda'), scores=torch.empty(N, 3, device='cuda'))
def run():
    L.check(e.lib.mcg_decoder_forward(e._handle, _stream(), tab, N, T, H, W, None, _ptr(out['gaze']), _ptr(out['boxes']), _ptr(out['scores']), _ptr(ws), ws.numel()), 'dec')
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): run()
torch.cuda.synchronize(); print(f'decoder only ({prec}, {clips} clips, {" ".join(sys.argv[4:]) or "default options"}): {(time.perf_counter() - t0) / iters * 1e3:.3f} ms/step')
