"""LAB (CPU): which operand precisions could replace f16x3, layer class by layer class?  
The fp32 CPU oracle is re-run on the parity-fuzz population (synth.fuzz_case) with the operands of every contraction of a layer CLASS
rounded to the significand a candidate arithmetic keeps -- 22 bits (fp16 high + low: what f16x3 feeds the matrix pipe), 11 bits (one
fp16: the '2-MFMA' forms x11 x w22 and x22 x w11) -- f32 accumulation, everything that is not a contraction in f32 like the engines.
Classes: stem, bb1x1 (backbone 1x1), bb3x3 (backbone 3x3), lat (FPN laterals), fpn3x3 (FPN output convs), dec (every decoder linear / bmm).
Per candidate: inputs within 1e-3 rad on (yaw, pitch), the worst, and the angle between gaze vectors (median / max).
usage: emulate_arith.py [cases=100] [seed=2] [family=uniform|trained] [threads=8]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as RealF
from mcgaze_amd import synth
from oracle import mcgaze_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
family = sys.argv[3] if len(sys.argv) > 3 else 'uniform'
torch.set_num_threads(int(sys.argv[4]) if len(sys.argv) > 4 else 8)


def p11(t): return t.half().float()
def p22(t):
    hi = t.half().float()
    return hi + (t - hi).half().float()
def p32(t): return t
P = {11: p11, 22: p22, 32: p32}

CLASSES = ('stem', 'bb1x1', 'bb3x3', 'lat', 'fpn3x3', 'dec')
def cand(default, **over):
    d = {c: default for c in CLASSES}
    d.update(over)
    return d
F16X3, A11, W11 = (22, 22), (11, 22), (22, 11)
CANDIDATES = [
    ('f16x3 everywhere (3 MFMAs per product; validates the emulation: the engine measures ~1e-5 rad)', cand(F16X3)),
    ('x11 x w22 everywhere (2 MFMAs: fp16 activations, split weights)', cand(A11)),
    ('x22 x w11 everywhere (2 MFMAs: split activations, fp16 weights)', cand(W11)),
    ('x11 x w22 on the FPN 3x3 convs only (46 % of the FLOPs), f16x3 elsewhere', cand(F16X3, fpn3x3=A11)),
    ('x22 x w11 on the FPN 3x3 convs only, f16x3 elsewhere', cand(F16X3, fpn3x3=W11)),
    ('x11 x w22 on every 3x3 conv (FPN + backbone), f16x3 elsewhere', cand(F16X3, fpn3x3=A11, bb3x3=A11)),
    ('x11 x w22 on the whole trunk, f16x3 decoder', cand(A11, dec=F16X3)),
    ('x11 x w22 on the decoder only, f16x3 trunk', cand(F16X3, dec=A11)),
    ('x11 x w22 on the backbone only (stem + layer1-4), f16x3 FPN + decoder', cand(F16X3, stem=A11, bb1x1=A11, bb3x3=A11)),
]

region = ['dec']
mode = [None]
def shim():
    F = types.SimpleNamespace(**{k: getattr(RealF, k) for k in dir(RealF) if not k.startswith('_')})
    def conv2d(x, w, b=None, **kw):
        k = w.shape[-1]
        if region[0] == 'bb':
            cls = 'stem' if k == 7 else ('bb3x3' if k == 3 else 'bb1x1')
        elif region[0] == 'fpn':
            cls = 'fpn3x3' if k == 3 else 'lat'
        else:
            cls = 'dec'
        xb, wb = mode[0][cls]
        return RealF.conv2d(P[xb](x), P[wb](w), b, **kw)
    def linear(x, w, b=None):
        xb, wb = mode[0]['dec']
        return RealF.linear(P[xb](x), P[wb](w), b)
    F.conv2d, F.linear = conv2d, linear
    T_ = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith('__')})
    def bmm(a, b):
        xb, wb = mode[0]['dec']
        return torch.bmm(P[xb](a), P[wb](b))
    T_.bmm = bmm
    return F, T_

real_resnet, real_fpn = orc.resnet, orc.fpn
def resnet(*a, **k):
    region[0] = 'bb'
    try: return real_resnet(*a, **k)
    finally: region[0] = 'dec'
def fpn(*a, **k):
    region[0] = 'fpn'
    try: return real_fpn(*a, **k)
    finally: region[0] = 'dec'

ks = [synth.fuzz_case(seed, c) for c in range(cases)]
sds = {}
refs = []
t0 = time.time()
for k in ks:
    if k['wseed'] not in sds:
        sds[k['wseed']] = synth.make_state_dict(k['wseed'], family=family)
    _, ref = orc.forward(sds[k['wseed']], k['img'], k['metas'], k['T'])
    refs.append(ref['gaze_score'])
print(f'# {cases} fuzz inputs (seed {seed}, weight family {family}); fp32 oracle pass {time.time() - t0:.0f} s', flush=True)
print('| candidate | within 1e-3 rad on (yaw, pitch) | worst d(yaw, pitch) | angle between gaze vectors: median / max |')
print('|---|---|---|---|', flush=True)
Fs, Ts = shim()
for name, m in CANDIDATES:
    mode[0] = m
    orc.F, orc.torch, orc.resnet, orc.fpn = Fs, Ts, resnet, fpn
    try:
        devs, angs = [], []
        for k, ref in zip(ks, refs):
            _, out = orc.forward(sds[k['wseed']], k['img'], k['metas'], k['T'])
            g = out['gaze_score']
            devs.append(float(orc.wrap_yaw(orc.yaw_pitch(g) - orc.yaw_pitch(ref)).abs().max()))
            angs.append(float((2 * torch.asin(((g.double() - ref.double()).norm(dim=-1) / 2).clamp(max=1))).max()))
    finally:
        orc.F, orc.torch, orc.resnet, orc.fpn = RealF, torch, real_resnet, real_fpn
    devs, angs = np.array(devs), np.array(angs)
    print(f'| {name} | {int((devs <= 1e-3).sum())} / {cases} | {devs.max():.2e} | {np.median(angs):.2e} / {angs.max():.2e} |', flush=True)
