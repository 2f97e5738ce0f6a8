"""LAB (CPU): what would a 16-bit-STORAGE engine deviate by?  The fp32 CPU oracle is re-run with every contraction's operands and
result rounded to bf16 (the existing throughput engine: validates the emulation against its measured deviation) or fp16 (the
engine DESIGN.md 8.3 sketches, not built), f32 accumulate in between, LayerNorm / softmax / boxes / RoIAlign in f32 like the engine.
Prints, per mode, the angle between gaze vectors and the oracle's, max |d(yaw, pitch)|, and the MAE shift against a synthetic
ground truth ~10.7 degrees from the oracle (bench.py mae_proxy's quantity, here per frame without windows).
usage: python tools/lab/emulate_16bit.py [clips=24]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as RealF
from mcgaze_amd import synth, metric
from oracle import mcgaze_oracle as orc

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 24
torch.set_num_threads(16)
T = 7


def shim(q):
    r = (lambda t: t) if q is None else (lambda t: t.to(q).float())
    F = types.SimpleNamespace(**{k: getattr(RealF, k) for k in dir(RealF) if not k.startswith('_')})
    F.conv2d = lambda x, w, b=None, **kw: r(RealF.conv2d(r(x), r(w), b, **kw))
    F.linear = lambda x, w, b=None: r(RealF.linear(r(x), r(w), b))
    return F, (lambda a, b: r(torch_bmm(r(a), r(b))))


torch_bmm = torch.bmm
sd = synth.make_state_dict(0)
metas = synth.make_img_metas(T)
out = {}
for name, q in (('fp32', None), ('bf16', torch.bfloat16), ('fp16', torch.float16)):
    F, bmm = shim(q)
    orc.F = F
    orc.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith('__')})
    orc.torch.bmm = bmm
    g = []
    for c in range(clips):
        _, ref = orc.forward(sd, synth.make_clips(9000 + c, 1, T), metas, T)
        g.append(ref['gaze_score'])
    out[name] = torch.cat(g)
    orc.F, orc.torch = RealF, torch
ref = out['fp32']
rs = np.random.RandomState(7)
noisy = ref + 0.155 * torch.from_numpy(rs.standard_normal(tuple(ref.shape)).astype(np.float32))
gt = noisy / noisy.norm(dim=1, keepdim=True)
base = float(metric.compute_angular_error(ref, gt))
for name in ('bf16', 'fp16'):
    g = out[name]
    ang = 2 * torch.asin(((g.double() - ref.double()).norm(dim=-1) / 2).clamp(max=1))
    d = orc.yaw_pitch_diff(g, ref).max(dim=1).values
    sh = float(metric.compute_angular_error(g, gt)) - base
    per_clip = d.reshape(clips, T).max(dim=1).values
    print(f'{name}: angle vs fp32 oracle median {float(ang.median()):.2e} mean {float(torch.rad2deg(ang).mean()):.3f} deg max {float(ang.max()):.2e} rad; '
          f'max |d(yaw,pitch)| per clip: median {float(per_clip.median()):.2e} max {float(per_clip.max()):.2e}; clips within 1e-3: {int((per_clip < 1e-3).sum())}/{clips}; '
          f'MAE shift vs synthetic gt ({base:.2f} deg): {sh:+.4f} deg')
