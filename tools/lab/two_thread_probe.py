"""GPU tool: localise a failure of tests/test_gpu_forward.py::test_two_threads_two_engines_one_device.  Two engines, two threads, one
device, repeated; per configuration (engine options switched off one at a time) the number of trials with any bitwise mismatch against
the single-threaded results, and what differed.  usage: python tools/lab/two_thread_probe.py [trials=6] [precision=f16x3]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcgaze_amd import synth
from mcgaze_amd.engine import HipEngine

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
prec = sys.argv[2] if len(sys.argv) > 2 else 'f16x3'
sd = synth.make_state_dict(0)
T, B = 7, 10
imgs = [[torch.from_numpy(synth.make_clips(500 + 10 * t + i, B, T)).to('cuda:0') for i in range(4)] for t in range(2)]
if len(sys.argv) > 3 and sys.argv[3] == 'full':
    CONFIGS = [('default', {}), ('trunk_streams=1', {'trunk_streams': 1}), ('bottleneck_fused=0', {'bottleneck_fused': 0}),
               ('pointwise_stream=0', {'pointwise_stream': 0}), ('decoder_chain=0', {'decoder_chain': 0}), ('stem_fused=0', {'stem_fused': 0}),
               ('same stream pool off: one thread only', {'_single': 1})]
    for name, opts in CONFIGS:
        engines = [HipEngine(sd, precision=prec) for _ in range(2)]
        for e in engines:
            for k, v in opts.items():
                if not k.startswith('_'):
                    e.set_option(k, v)
        want = [[{k: v.clone() for k, v in engines[0].forward(x, T).items()} for x in imgs[t]] for t in range(2)]
        torch.cuda.synchronize()
        bad, notes = 0, []
        for trial in range(trials):
            got = [[None] * 8 for _ in range(2)]

            def work(t):
                s = torch.cuda.Stream(device='cuda:0')
                with torch.cuda.stream(s):
                    for r in range(8):
                        got[t][r] = {k: v.clone() for k, v in engines[t].forward(imgs[t][r % 4], T).items()}
                s.synchronize()
            if opts.get('_single'):
                work(0); work(1)
            else:
                th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
            torch.cuda.synchronize()
            miss = []
            for t in range(2):
                for r in range(8):
                    for k in ('gaze', 'boxes', 'scores'):
                        if not torch.equal(got[t][r][k], want[t][r % 4][k]):
                            d = (got[t][r][k] - want[t][r % 4][k]).abs()
                            nz = d.reshape(d.shape[0] if k != 'gaze' else d.shape[1], -1) if k != 'gaze' else d.permute(1, 0, 2).reshape(d.shape[1], -1)
                            frames = torch.nonzero(nz.amax(dim=1) > 0).flatten().tolist()
                            miss.append(f't{t} r{r} {k}: max {float(d.max()):.2e}, {len(frames)} frames differ (first {frames[:6]})')
            if miss:
                bad += 1
                notes.append(miss[:4])
        print(f'{name}: {bad} of {trials} trials with a mismatch', flush=True)
        for n_ in notes[:3]:
            for m in n_:
                print('    ' + m)


def trunk_probe(label, opts, ntr, levels_of, names):
    engines = [HipEngine(sd, precision=prec) for _ in range(2)]
    for e in engines:
        for k, v in opts.items():
            e.set_option(k, v)
    wantp = [[[p.clone() for p in levels_of(engines[0], x)] for x in imgs[t]] for t in range(2)]
    torch.cuda.synchronize()
    bad = 0
    for trial in range(ntr):
        gotp = [[None] * 8 for _ in range(2)]

        def workp(t):
            s = torch.cuda.Stream(device='cuda:0')
            with torch.cuda.stream(s):
                for r in range(8):
                    gotp[t][r] = [p.clone() for p in levels_of(engines[t], imgs[t][r % 4])]
            s.synchronize()
        th = [threading.Thread(target=workp, args=(t,)) for t in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        lines = []
        for t in range(2):
            for r in range(8):
                for lv in range(len(names)):
                    g, w = gotp[t][r][lv], wantp[t][r % 4][lv]
                    if not torch.equal(g, w):
                        d = (g - w).abs()
                        idx = torch.nonzero(d.amax(dim=3) > 0)
                        fr = sorted(set(idx[:, 0].tolist()))
                        ys, xs = idx[:, 1], idx[:, 2]
                        ch = torch.nonzero(d.amax(dim=(0, 1, 2)) > 0).flatten().tolist()
                        lines.append(f'  trial {trial} t{t} r{r} {names[lv]}: {idx.shape[0]} pixels, max {float(d.max()):.2e} (scale {float(w.abs().max()):.2f}); frames {fr[:8]}; '
                                     f'rows {int(ys.min())}..{int(ys.max())}, cols {int(xs.min())}..{int(xs.max())}; {len(ch)} channels (first {ch[:6]})')
        if lines:
            bad += 1
            for ln in lines[:6]:
                print(ln, flush=True)
    print(f'{label}: {bad} of {ntr} trials with a mismatch', flush=True)


PYR = (lambda e, x: e.backbone_fpn(x), ['P2', 'P3', 'P4', 'P5'])
LEV = (lambda e, x: e.backbone_only(x, return_levels=True), ['C2', 'C3', 'C4', 'C5'])
trunk_probe('C2..C5, trunk_streams=1', {'trunk_streams': 1}, 3 * trials, *LEV)
trunk_probe('C2..C5, trunk_streams=1, bottleneck_fused=0', {'trunk_streams': 1, 'bottleneck_fused': 0}, 3 * trials, *LEV)
trunk_probe('C2..C5, trunk_streams=1, pointwise_stream=0', {'trunk_streams': 1, 'pointwise_stream': 0}, 3 * trials, *LEV)
trunk_probe('pyramids, default', {}, 2 * trials, *PYR)
