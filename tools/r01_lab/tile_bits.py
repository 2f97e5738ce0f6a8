"""GPU tool: output of one conv shape under the MCG_TILE of the environment, saved for a bitwise comparison between tiles.
usage: MCG_TILE=t python tools/tile_bits.py out.pt N H W Cin Cout k stride pad ; python tools/tile_bits.py --cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == '--cmp':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for i, (x, y) in enumerate(zip(a, b)):
        d = (x.view(torch.int16) != y.view(torch.int16))
        print(f'run {i}: {int(d.sum())} differing elements of {d.numel()}', ('max abs diff %.3g' % float((x.float() - y.float()).abs().max())) if d.any() else '')
    sys.exit(0)
from mcgaze_amd import engine as E
out = sys.argv[1]
N, H, W, Cin, Cout, k, stride, pad = [int(v) for v in sys.argv[2:10]]
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(N, H, W, Cin, device='cuda', generator=g).to(torch.bfloat16)
w = (torch.randn(Cout, k, k, Cin, device='cuda', generator=g) / (Cin * k * k) ** 0.5).to(torch.bfloat16)
b = torch.randn(Cout, device='cuda', generator=g)
ys = []
for _ in range(4):
    ys.append(E.conv2d(x, w, b, stride=stride, pad=pad, relu=True).cpu())
torch.save(ys, out)
print('self-consistent:', all(torch.equal(ys[0].view(torch.int16), y.view(torch.int16)) for y in ys))
