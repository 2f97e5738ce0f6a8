"""GPU probe: the DynamicConv parameter generator (linear 1344 x 32768 x 256, 88 MB bf16 out) against a plain 88 MB memset:
the kernel is 4x off the write rate, i.e. bound by per-workgroup prologue / epilogue latency, not by HBM."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from mcgaze_amd import engine as E
x = torch.empty(1344, 32768, dtype=torch.bfloat16, device='cuda')
for _ in range(5): x.zero_()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): x.zero_()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f'memset 88 MB: {dt*1e6:.1f} us  {x.numel()*2/dt/1e12:.2f} TB/s')
a = torch.randn(1344, 1, 1, 256, device='cuda').to(torch.bfloat16)
w = torch.randn(32768, 1, 1, 256, device='cuda').to(torch.bfloat16) * 0.06
b = torch.randn(32768, device='cuda')
for t in os.environ.get('TILES', '9 12 11 3').split():
    os.environ['MCG_TILE'] = t
    for _ in range(5): y = E.conv2d(a, w, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): y = E.conv2d(a, w, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f'tile {t}: linear 1344x32768x256: {dt*1e6:.1f} us  write {y.numel()*2/dt/1e12:.2f} TB/s')
