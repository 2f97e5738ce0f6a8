"""GPU tool: single-clip latency through the reference's call surface -- model(return_loss=False, rescale=True, img=[clip],
img_metas=[metas]) -- against the bare engine call, per precision.  usage: python tools/lab/api_latency.py [calls=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import init_detector, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
img = torch.from_numpy(synth.make_clips(5, 1, 7)).cuda()
metas = synth.make_img_metas(7, (224, 224, 3))
for prec in ('f16x3', 'bf16', 'fp32'):
    model = init_detector(os.path.join(root, 'configs', 'mcgaze', 'r50_clip7_gaze360.py'), None, device='cuda:0', precision=prec)
    eng = model.engine()
    def api():
        (_, _), g = model(return_loss=False, rescale=True, format=False, img=[img], img_metas=[metas])
        return g['gaze_score']
    def bare():
        return eng.forward(img, 7)['gaze']
    for name, fn in (('api', api), ('engine', bare)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn().cpu()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f'{prec:7s} {name:7s} median {ts[n // 2] * 1e3:.3f} ms  min {ts[0] * 1e3:.3f} ms', flush=True)
