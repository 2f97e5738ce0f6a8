import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for mb in (180, 719, 2876):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device='cuda').to(torch.bfloat16); r = torch.randn(n, device='cuda').to(torch.bfloat16); y = torch.empty_like(x)
    a = t(lambda: y.copy_(x)); b = t(lambda: torch.add(x, r, out=y)); c = t(lambda: x.sum())
    print(f'{mb} MB: copy {2*mb/1024/a/1e3:.2f} TB/s  add(2r+1w) {3*mb/1024/b/1e3:.2f} TB/s  read-only sum {mb/1024/c/1e3:.2f} TB/s')
y = torch.empty(719 * 1024 * 1024 // 2, device='cuda', dtype=torch.bfloat16)
a = t(lambda: y.fill_(1.0)); print(f'fill 719MB: {719/1024/a/1e3:.2f} TB/s')
x = torch.randn(180 * 1024 * 1024 // 2, device='cuda').to(torch.bfloat16)
a = t(lambda: torch.cat([x, x, x, x], out=y)); print(f'read 180MB x4 -> write 719MB: {719/1024/a/1e3:.2f} TB/s write-side')
