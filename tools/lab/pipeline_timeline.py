"""GPU tool: timeline of the two-stream batch pipeline -- when each batch's trunk and decoder start and end (HIP events on their
own streams).  Tells whether the decoder of batch k is hidden under the trunk of batch k+1 or sits on the critical path.
usage: python tools/lab/pipeline_timeline.py [steps]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import synth, lib as L
from mcgaze_amd.engine import HipEngine, PipelinedRunner, _ptr
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
PREC = next((a for a in sys.argv[1:] if a in ('f16x3', 'bf16', 'fp32')), 'f16x3')   # the product engine unless another is named
e = HipEngine(synth.make_state_dict(0), precision=PREC)
B, T = 64, 7
img = torch.from_numpy(synth.make_clips(3, B, T)).cuda()
r = PipelinedRunner(e, B * T, 224, 224, T)
out = [dict(gaze=torch.empty(4, B * T, 3, device='cuda'), boxes=torch.empty(B * T, 3, 4, device='cuda'), scores=torch.empty(B * T, 3, device='cuda')) for _ in range(2)]
hw = None
ev = lambda: torch.cuda.Event(enable_timing=True)
lib, h = e.lib, e._handle
for rnd in range(2):
    rec = []
    t0 = ev(); t0.record(r.sa)
    for k in range(K):
        slot = k & 1
        if r.used[slot]: r.sa.wait_event(r.dec_done[slot])
        a, b, c, d = ev(), ev(), ev(), ev()
        a.record(r.sa)
        L.check(lib.mcg_backbone_fpn_forward(h, C.c_void_p(r.sa.cuda_stream), _ptr(img), r.N, r.H, r.W, 0, r.tabs[slot], _ptr(r.trunk_ws), r.trunk_ws.numel()), 'trunk')
        b.record(r.sa); r.trunk_done[slot].record(r.sa)
        r.sb.wait_event(r.trunk_done[slot])
        c.record(r.sb)
        L.check(lib.mcg_decoder_forward(h, C.c_void_p(r.sb.cuda_stream), r.tabs[slot], r.N, T, 224, 224, _ptr(hw), _ptr(out[slot]['gaze']), _ptr(out[slot]['boxes']),
                                        _ptr(out[slot]['scores']), _ptr(r.dec_ws), r.dec_ws.numel()), 'decoder')
        d.record(r.sb); r.dec_done[slot].record(r.sb); r.used[slot] = True
        rec.append((a, b, c, d))
    torch.cuda.synchronize()
    if rnd == 0: continue
    print(' k  trunk start    end | decoder start    end | trunk ms  decoder ms')
    for k, (a, b, c, d) in enumerate(rec):
        ta, tb, tc, td = [t0.elapsed_time(x) for x in (a, b, c, d)]
        print(f'{k:2d}  {ta:8.3f} {tb:8.3f} | {tc:8.3f} {td:8.3f} | {tb - ta:7.3f}  {td - tc:7.3f}')
    print(f'steady-state step: {(t0.elapsed_time(rec[-1][1]) - t0.elapsed_time(rec[1][1])) / (K - 2):.3f} ms')
