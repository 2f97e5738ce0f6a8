"""GPU probe: does the ORDER in which two engines are created and timed inside one process change their rates?
(bench.py times the bf16 engine and then the f16x3 engine; the second leg measured 12 % slower than alone.)
usage: python tools/lab/leg_order_probe.py <order> [steps]   order = e.g. x3,x3 | bf16,x3 | x3,bf16 | bf16,del,x3"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from mcgaze_amd import synth

order = sys.argv[1].split(',')
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
argv, sys.argv = sys.argv, ['bench.py']
a = bench.parse()
sys.argv = argv
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
img = torch.from_numpy(synth.make_clips(3, 64, 7)).to(dev)
legs = []
for name in order:
    if name == 'del':
        legs.clear(); gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
        continue
    prec = {'x3': 'f16x3'}.get(name, name)
    leg = bench.Leg(a, prec, dev, 1, 0, None, img, 64, 7)
    legs.append(leg)
    t = leg.timed(steps, 5)
    print(f'{name}: {64 * steps / t:.1f} clips/s  {t / steps * 1e3:.3f} ms/step', flush=True)
