"""GPU probe: where does the bf16 engine's deviation come from?  For fuzz cases: (a) the bf16 engine, (b) bf16 trunk (pyramid converted to f32)
+ f16x3 decoder, (c) f16x3 trunk (pyramid rounded to bf16) + bf16 decoder; angle between gaze vectors vs the oracle.
usage: python tools/lab/mixed_precision_probe.py [cases=40] [seed=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mcgaze_amd import synth, engine as E
from mcgaze_amd.engine import HipEngine
from oracle import mcgaze_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.set_num_threads(16)
sds, engs = {}, {}
res = {k: [] for k in ('bf16', 'bf16 trunk + x3 decoder', 'x3 trunk + bf16 decoder', 'f16x3', 'x3 trunk + fp32 decoder', 'fp32 trunk + x3 decoder', 'fp32')}

def decode(eng, split, pyr, metas, sd, T):
    boxes, obj = orc.init_proposals(orc.as_torch(sd), metas)
    dt = pyr[0].dtype
    b, o = boxes.cuda().contiguous(), obj.to(dt).cuda().contiguous()
    for s in range(4):
        roi, _ = E.roi_align(pyr, b)
        o, b, _ = E.stage_forward(eng.weights.stages[s], roi, o, b, T, split=split)
    return E.gaze_head(eng.weights.gaze, o, split=split)[0].float().cpu()

def ang(a, b):
    return float((2 * torch.asin(((a.double() - b.double()).norm(dim=-1) / 2).clamp(max=1))).max())

for c in range(cases):
    k = synth.fuzz_case(seed, c)
    w = k['wseed']
    if w not in sds:
        sds[w] = synth.make_state_dict(w)
        engs[w] = {p: HipEngine(sds[w], precision=p) for p in ('bf16', 'f16x3', 'fp32')}
    _, ref = orc.forward(sds[w], k['img'], k['metas'], k['T'])
    x = torch.from_numpy(k['img']).cuda()
    eb, ex, ef = engs[w]['bf16'], engs[w]['f16x3'], engs[w]['fp32']
    pb, px, pf = eb.backbone_fpn(x), ex.backbone_fpn(x), ef.backbone_fpn(x)
    g = {'bf16': decode(eb, False, pb, k['metas'], sds[w], k['T']),
         'bf16 trunk + x3 decoder': decode(ex, True, [p.float() for p in pb], k['metas'], sds[w], k['T']),
         'x3 trunk + bf16 decoder': decode(eb, False, [p.to(torch.bfloat16) for p in px], k['metas'], sds[w], k['T']),
         'f16x3': decode(ex, True, px, k['metas'], sds[w], k['T']),
         'x3 trunk + fp32 decoder': decode(ef, False, px, k['metas'], sds[w], k['T']),
         'fp32 trunk + x3 decoder': decode(ex, True, pf, k['metas'], sds[w], k['T']),
         'fp32': decode(ef, False, pf, k['metas'], sds[w], k['T'])}
    for name, v in g.items():
        res[name].append(ang(v, ref['gaze_score']))
for name, v in res.items():
    v = np.array(v)
    print(f'{name:28s} angle vs oracle: median {np.median(v):.2e}  p90 {np.percentile(v, 90):.2e}  max {v.max():.2e} rad')
