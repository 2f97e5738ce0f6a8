"""GPU tool: which launches of the f16x3 step run ON the package power cap and which below it.  Each operator class of the step is looped
stand-alone for ~2 s under rocm-smi sampling: ms, shader clock, package power, energy per launch.
usage: power_map.py [frames=448] [seconds=2.0]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _smi import sampled
from mcgaze_amd import engine as E, lib as L, synth
from mcgaze_amd.packing import bneck_stream, split_pack, wino_pack
N = int(sys.argv[1]) if len(sys.argv) > 1 else 448
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
lib = L.load()
s = E._stream()
g = torch.Generator().manual_seed(1)
rows = []
def report(name, fn, flops=0.0, gbytes=0.0):
    ms, clk, pw, n = sampled(fn, secs)
    rows.append((name, ms, clk, pw))
    print(f'| {name} | {ms:.3f} | {clk:.0f} | {pw:.0f} | {ms * pw / 1e3:.2f} | {flops / ms / 1e9 if flops else 0:.0f} | {gbytes / ms * 1e3 if gbytes else 0:.0f} |', flush=True)
print('| launch (frames = %d) | ms | sclk MHz | W | J / launch | algorithmic TF/s | algorithmic GB/s |\n|---|---|---|---|---|---|---|' % N)

def conv(name, H, W, Cin, Cout, k, stride, relu=True, data='relu', res=False):
    x = torch.randn(N, H, W, Cin, device='cuda')
    if data == 'relu': x.relu_()
    w = torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin / 2) ** 0.5
    ws = split_pack(w.reshape(Cout, -1)).cuda()
    b = torch.randn(Cout, device='cuda')
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    y = torch.empty(N, Ho, Wo, Cout, device='cuda')
    r = torch.randn(N, Ho, Wo, Cout, device='cuda').relu_() if res else None
    d = L.ConvDesc(x.data_ptr(), ws.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), N, H, W, Cin, Cout, k, k, stride, k // 2, int(relu), 1 if res else 0, 0, 0, None, 0, 1, 0, 0, 0, 0)
    fl = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    gb = 4.0 * N * (H * W * Cin + Ho * Wo * Cout * (2 if res else 1)) / 1e9
    report(name, lambda: L.check(lib.mcg_conv2d(s, L.MCG_F16X3, C.byref(d)), name), fl, gb)

def wino(name, H, W, Cin, Cout, tile, data='randn'):
    x = torch.randn(N, H, W, Cin, device='cuda')
    if data == 'relu': x.relu_()
    w = torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5
    u = wino_pack(w, g=2).cuda()
    b = torch.randn(Cout, device='cuda')
    y = torch.empty(N, H, W, Cout, device='cuda')
    fl = 2.0 * N * H * W * Cout * Cin * 9
    report(name, lambda: L.check(lib.mcg_conv3x3_wino_x3(s, E._ptr(x), E._ptr(u), E._ptr(b), E._ptr(y), N, H, W, Cin, Cout, 1, tile, 0.0, 2), name), fl, 4.0 * N * H * W * (Cin + Cout) / 1e9)

def bneck(name, H, W, cm, nsrc, cn):
    w2 = torch.randn(cm, 3, 3, cm, generator=g) / (9 * cm / 2) ** 0.5
    w3 = torch.randn(4 * cm, cm + 64 * (nsrc - 1), generator=g) / 8
    w1 = torch.randn(cn, 4 * cm, generator=g) / 11 if cn else None
    ws, bs = bneck_stream(w2, torch.randn(cm, generator=g) * 0.1, w3, torch.randn(4 * cm, generator=g) * 0.1, w1, torch.randn(cn, generator=g) * 0.1 if cn else None)
    ws, bs = ws.cuda(), bs.cuda()
    x = torch.randn(N, H, W, cm, device='cuda').relu_()
    src2 = torch.randn(N, H, W, 4 * cm if nsrc == 1 else 64, device='cuda').relu_()
    M = N * H * W
    gb = 4.0 * M * (cm + (4 * cm if nsrc == 1 else 64) + 4 * cm + cn) / 1e9
    fl = 2.0 * M * (9 * cm * cm + (cm + 64 * (nsrc - 1)) * 4 * cm + 4 * cm * cn)
    report(name, lambda: E.bottleneck_x3(x, src2, ws, bs, cn, nsrc), fl, gb)

wino('FPN P2 3x3 256->256 56x56, wino 8-wave tile', 56, 56, 256, 256, 1)
wino('FPN P2 3x3, wino one-wave-per-SIMD tile', 56, 56, 256, 256, 4)
conv('FPN P2 3x3 direct x3', 56, 56, 256, 256, 3, 1, relu=False, data='randn')
wino('layer3 conv2 256->256 14x14 wino 8-wave (relu input)', 14, 14, 256, 256, 1, 'relu')
wino('layer3 conv2 wino one-wave-per-SIMD', 14, 14, 256, 256, 4, 'relu')
conv('layer3 conv1 1x1 1024->256 14x14', 14, 14, 1024, 256, 1, 1)
conv('layer3 conv3 1x1 256->1024 + res 14x14 (pw_single_x3)', 14, 14, 256, 1024, 1, 1, res=True)
conv('layer4 conv1 1x1 2048->512 7x7', 7, 7, 2048, 512, 1, 1)
conv('layer4 conv3 1x1 512->2048 + res 7x7', 7, 7, 512, 2048, 1, 1, res=True)
conv('layer3.b0 conv2 3x3/2 256->256 28->14', 28, 28, 256, 256, 3, 2)
conv('lateral P2 1x1 256->256 56x56', 56, 56, 256, 256, 1, 1, relu=False)
bneck('fused tail layer1 identity (cm 64, cn 64)', 56, 56, 64, 1, 64)
bneck('fused tail layer1.b2 (cm 64, cn 128)', 56, 56, 64, 1, 128)
bneck('fused tail layer2 identity (cm 128, cn 128)', 28, 28, 128, 1, 128)
# whole engine legs
from mcgaze_amd.engine import HipEngine
eng = HipEngine(synth.make_state_dict(0), precision='f16x3')
img = torch.from_numpy(synth.make_clips(3, N // 7, 7)).cuda()
report('whole forward (trunk + decoder, one stream pair)', lambda: eng.forward(img, 7), 99.55e9 * (N // 7))
report('trunk only (backbone + FPN, two frame ranges)', lambda: eng.backbone_fpn(img), 97.01e9 * (N // 7))
# round 6: the decoder alone on precomputed pyramids, and the 16-bit throughput engines' whole forward
from mcgaze_amd.engine import _ptr, _ws, _stream
pyr = eng.backbone_fpn(img)
tab = (C.c_void_p * 4)(*[p.data_ptr() for p in pyr])
ws = _ws(eng.lib.mcg_decoder_workspace_bytes(eng._handle, N), eng.device)
out = dict(gaze=torch.empty(4, N, 3, device='cuda'), boxes=torch.empty(N, 3, 4, device='cuda'), scores=torch.empty(N, 3, device='cuda'))
report('decoder only (4 x [RoIAlign + stage] + gaze head)', lambda: L.check(eng.lib.mcg_decoder_forward(eng._handle, _stream(), tab, N, 7, 224, 224, None, _ptr(out['gaze']), _ptr(out['boxes']), _ptr(out['scores']), _ptr(ws), ws.numel()), 'dec'), 2.54e9 * (N // 7))
del eng, pyr, ws
for prec in ('f16', 'bf16'):
    e16 = HipEngine(synth.make_state_dict(0), precision=prec)
    report(f'whole forward, {prec} engine', lambda: e16.forward(img, 7), 99.55e9 * (N // 7))
    del e16
