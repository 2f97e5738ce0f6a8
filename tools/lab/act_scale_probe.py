"""GPU tool: the f16x3 contraction's error against the f64 product as a function of the ACTIVATIONS' magnitude (1x1 conv, K = 256), next to the f32
kernel: below |x| ~ 0.1 the fp16 low halves of the activations are subnormal (tests/test_gpu_kernels.py::test_f16x3_activation_magnitudes).
usage: python tools/lab/act_scale_probe.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from mcgaze_amd import engine as E
def scale_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
g = torch.Generator().manual_seed(99)
N, H, W, Cin, Cout = 2, 12, 12, 256, 256
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to('cuda:0')
for kind in ('randn', 'relu'):
    for xs in (1e2, 1.0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
        x = torch.randn(N, Cin, H, W, generator=g)
        if kind == 'relu': x = x.relu()
        x = x * xs
        w = torch.randn(Cout, Cin, 1, 1, generator=g) / 16
        ref = F.conv2d(x.double(), w.double())
        y = E.conv2d(nhwc(x), nhwc(w), None, split=True)
        y32 = E.conv2d(nhwc(x), nhwc(w), None)   # fp32 engine path
        torch.cuda.synchronize()
        print(f'{kind} activations ~{xs:g}: f16x3 {scale_err(y.permute(0,3,1,2), ref):.2e} of scale, fp32 kernel {scale_err(y32.permute(0,3,1,2), ref):.2e}')
