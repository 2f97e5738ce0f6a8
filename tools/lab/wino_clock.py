"""GPU tool: one 3x3 conv shape through mcg_conv3x3_wino_x3 with a list of tile values, each looped for ~2.5 s while a thread samples
rocm-smi (sclk, package power): ms per launch, clock, watts, and the launch's duration in shader cycles (ms x clock) -- on a power-capped
chip a variant that saves cycles can run at a lower clock, and only the two numbers together say what a change did.
usage: wino_clock.py N H W Cin Cout tile[,tile...] [data randn|relu] [seconds]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import engine as E, lib as L
from mcgaze_amd.packing import wino_pack
N, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
tiles = [int(t) for t in sys.argv[6].split(',')]
mode = sys.argv[7] if len(sys.argv) > 7 else 'randn'
secs = float(sys.argv[8]) if len(sys.argv) > 8 else 2.5
lib = L.load()
x = torch.randn(N, H, W, Cin, device='cuda')
if mode == 'relu': x.relu_()
w = torch.randn(Cout, 3, 3, Cin) / (9 * Cin) ** 0.5
b = torch.randn(Cout, device='cuda')
u = wino_pack(w, g=2).cuda()
y = torch.empty(N, H, W, Cout, device='cuda')
s = E._stream()
def sample(out, stop):
    while not stop.is_set():
        t = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
        m = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', t); pw = re.search(r'Power \(W\): ([\d.]+)', t)
        if m and pw: out.append((int(m.group(1)), float(pw.group(1))))
        time.sleep(0.1)
fl = 2.0 * N * H * W * Cout * Cin * 9
for tile in tiles:
    def fn(): L.check(lib.mcg_conv3x3_wino_x3(s, E._ptr(x), E._ptr(u), E._ptr(b), E._ptr(y), N, H, W, Cin, Cout, 1, tile, 0.0, 2), 'wino')
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while n < 20 or time.perf_counter() - t0 < 0.3:
        fn(); n += 1
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / n
    iters = max(50, int(secs / est))
    out, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(out, stop)); 
    t0 = time.perf_counter()
    for i in range(iters):
        fn()
        if i == iters // 5: th.start()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    stop.set(); th.join()
    out = out[1:-1] if len(out) > 4 else out
    clk = sum(o[0] for o in out) / max(1, len(out)); pw = sum(o[1] for o in out) / max(1, len(out))
    print(f'tile {tile:4d} (dbg {tile >> 4:2d}): {ms:.4f} ms  {fl / ms / 1e9:6.1f} TF/s  sclk {clk:6.0f} MHz  {pw:6.0f} W  {ms * clk / 1e3:7.3f} Mcycles/launch  {ms * pw / 1e3:.3f} J/launch  ({len(out)} samples, {mode})', flush=True)
