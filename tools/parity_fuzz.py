"""GPU tool: parity fuzz of the engines against the CPU oracle on random shapes -- clip length, batch, frame size (multiples of 32),
per-frame img_shape inside the padded frame, weight seed -- to look for inputs where the parity-grade engines leave north_star's
1e-3 rad on (yaw, pitch).  usage: python tools/parity_fuzz.py [cases=24] [seed=0] [precisions=f16x3,fp32] [diagnose=1] [family=uniform|trained] [engine options, e.g. winograd=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mcgaze_amd import synth
from mcgaze_amd.engine import HipEngine
from oracle import mcgaze_oracle as orc
from tests import parity_tools as PT

diag = []
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
precs = (sys.argv[3] if len(sys.argv) > 3 else 'f16x3,fp32').split(',')
diagnose = (sys.argv[4] if len(sys.argv) > 4 else '1') != '0'
family = sys.argv[5] if len(sys.argv) > 5 else 'uniform'   # synth.make_state_dict's weight family
opts = [kv.partition('=') for kv in sys.argv[6:]]           # mcg_engine_set_option on every engine
torch.set_num_threads(16)
engines, sds = {}, {}
worst = {p: 0.0 for p in precs}
import collections
counts = {p: collections.Counter() for p in precs}
print('| case | weights | B | T | H x W | img_shape | ' + ' | '.join(f'{p}: max d(yaw, pitch) rad / max angle between gaze vectors rad' for p in precs) + ' |')
print('|---|---|---|---|---|---|' + '---|' * len(precs))
for c in range(cases):
    k = synth.fuzz_case(seed, c)
    wseed, B, T, H, W, (ih, iw), full, img, metas = k['wseed'], k['B'], k['T'], k['H'], k['W'], k['img_shape'], k['full'], k['img'], k['metas']
    N = B * T
    if wseed not in sds:
        sds[wseed] = synth.make_state_dict(wseed, family=family)
    stages = []
    _, ref = orc.forward(sds[wseed], img, metas, T, collect=stages)
    want = orc.yaw_pitch(ref['gaze_score'])
    devs, notes = [], []
    for p in precs:
        key = (p, wseed)
        if key not in engines:
            engines[key] = HipEngine(sds[wseed], precision=p)
            for name, _, val in opts:
                engines[key].set_option(name, int(val))
        hw = None if full else np.tile(np.array([ih, iw], dtype=np.int32), (N, 1))
        out = engines[key].forward(torch.from_numpy(img).cuda(), T, img_hw=hw)
        d = float(orc.wrap_yaw(orc.yaw_pitch(out['gaze'][0].cpu()) - want).abs().max())
        worst[p] = max(worst[p], d)
        ang = float((2 * torch.asin(((out['gaze'][0].cpu().double() - ref['gaze_score'].double()).norm(dim=-1) / 2).clamp(max=1))).max())
        devs.append(f'{d:.2e} / {ang:.2e}')
        if not np.isfinite(d):
            counts[p]['NOT FINITE'] += 1
        elif d > 1e-3 and not diagnose:
            counts[p]['beyond 1e-3'] += 1
        elif d > 1e-3:
            rep = PT.stage_report(engines[key], p, sds[wseed], img, metas, T, stages)
            note, disc = f'{p}: ' + PT.describe(rep), rep['discontinuity']
            err = orc.wrap_yaw(orc.yaw_pitch(out['gaze'][0].cpu()) - want)
            f = int(err.abs().max(dim=1).values.argmax())
            gy = float(ref['gaze_score'][f, 1])
            vec = float((out['gaze'][0].cpu()[f] - ref['gaze_score'][f]).norm())
            notes.append(note + f'; worst frame: oracle gaze y = {gy:+.4f} (yaw = atan2(x, -z) is ill-conditioned as |y| -> 1), |d gaze vector| = {vec:.2e}')
            counts[p]['discontinuity' if disc else 'ill-conditioned / amplified'] += 1
        else:
            counts[p]['within 1e-3'] += 1
    print(f'| {c} | {wseed} | {B} | {T} | {H} x {W} | {ih} x {iw} | ' + ' | '.join(devs) + ' |', flush=True)
    for n_ in notes:
        diag.append(f'case {c}: ' + n_)
print()
for line in diag:
    print(line)
print()
print('worst: ' + ', '.join(f'{p} {worst[p]:.3e} rad' for p in precs) + '  (north_star tolerance 1e-3)')
for p in precs:
    print(f'{p}: ' + ', '.join(f'{k}: {v}' for k, v in sorted(counts[p].items())) + f' of {cases} cases')
# machine-readable summary, merged over runs into gpurun_out/parity_fuzz.json (copy to profiles/parity_fuzz.json: bench.py quotes it as
# `parity_fuzz` with the library build id it was taken on)
import json
from mcgaze_amd import lib as L
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, 'gpurun_out', 'parity_fuzz.json')
os.makedirs(os.path.dirname(path), exist_ok=True)
doc = json.load(open(path)) if os.path.exists(path) else {}
if doc.get('build_id') != L.build_id() or doc.get('family') != family or doc.get('options') != [f'{n}={v}' for n, _, v in opts]:
    doc = {'build_id': L.build_id(), 'family': family, 'options': [f'{n}={v}' for n, _, v in opts], 'runs': {}}
doc['runs'][f'seed {seed}'] = {'cases': cases, 'engines': {p: {'within_1e-3': counts[p]['within 1e-3'], 'worst_rad': worst[p],
                                                              'beyond': {k: v for k, v in counts[p].items() if k != 'within 1e-3'}} for p in precs},
                               'exceptions': diag}
for p in precs:
    runs = [r for r in doc['runs'].values() if p in r['engines']]
    doc[p] = {'n': sum(r['cases'] for r in runs), 'within': sum(r['engines'][p]['within_1e-3'] for r in runs), 'worst_rad': max(r['engines'][p]['worst_rad'] for r in runs)}
json.dump(doc, open(path, 'w'), indent=1)
