"""GPU probe: where does the bf16 deviation come from?  Mixes fp32/bf16 trunk and decoder."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mcgaze_amd import synth, engine as E
from mcgaze_amd.packing import PackedWeights
from oracle import mcgaze_oracle as orc

sd = synth.make_state_dict(0)
g = dict(np.load('tests/golden/batch2.npz'))
img = torch.from_numpy(synth.make_clips(int(g['img_seed']), 2, 7)).cuda()
ref = torch.from_numpy(g['gaze_score'])
T = 7
eng = {p: E.HipEngine(sd, precision=p) for p in ('fp32', 'bf16')}
pw = {p: eng[p].weights for p in eng}

def decoder(pyr, p):
    dt = torch.float32 if p == 'fp32' else torch.bfloat16
    w = pw[p]
    pyr = [x.to(dt) for x in pyr]
    N = pyr[0].shape[0]
    e = sd['rpn_head.init_proposal_bboxes.weight']
    boxes, obj = orc.init_proposals(orc.as_torch(sd), synth.make_img_metas(N))
    boxes, obj = boxes.cuda(), obj.to(dt).cuda()
    for s in range(4):
        roi, _ = E.roi_align(pyr, boxes)
        obj, boxes, cls = E.stage_forward(w.stages[s], roi, obj, boxes, T)
    return E.gaze_head(w.gaze, obj)[0].cpu()

def report(name, gz):
    d = orc.yaw_pitch_diff(gz, ref)
    ang = torch.rad2deg(torch.acos((gz * ref).sum(-1).clamp(-1, 1)))
    print(f'{name:32s} max|d(yaw,pitch)| {d.max():.2e} rad   mean ang err {ang.mean():.3f} deg  max {ang.max():.3f}')

for tp in ('fp32', 'bf16'):
    pyr = eng[tp].backbone_fpn(img)
    for dp in ('fp32', 'bf16'):
        report(f'trunk {tp} + decoder {dp}', decoder(pyr, dp))
# timing, first look
for p, B in (('bf16', 64), ('bf16', 8), ('fp32', 8)):
    x = torch.from_numpy(synth.make_clips(3, B, 7)).cuda()
    for chunk in (0, 56, 28):
        eng[p].forward(x, 7, chunk_frames=chunk); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3): eng[p].forward(x, 7, chunk_frames=chunk)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 3
        print(f'{p} B={B} chunk={chunk}: {dt*1e3:.2f} ms/step  {B/dt:.1f} clips/s')
