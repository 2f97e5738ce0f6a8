"""GPU lab tool: where the fused bottleneck tail's time and JOULES go.  Each variant library (tools/lab/bneck_ablate.sh: one memory stream
of bneck_x3_kernel switched off) runs the same three block shapes in a child process (MCGAZE_LIB selects the library) for ~2 s each under
rocm-smi sampling: ms, shader clock, package power, joules per launch.  Differences against the product library price each stream.
usage: bneck_ablate.py [frames=448] [seconds=2.0]           (child mode: bneck_ablate.py --child frames seconds)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
SHAPES = [('layer1 identity (cm 64, cn 64, 56x56)', 56, 64, 64), ('layer1.b2 (cm 64, cn 128, 56x56)', 56, 64, 128), ('layer2 identity (cm 128, cn 128, 28x28)', 28, 128, 128)]
NAMES = {0: 'product', 1: 'no weight stream (ring filled once)', 2: 'no residual loads', 4: 'no y / z stores', 8: 'no window loads', 15: 'no memory stream at all'}


def child(N, secs):
    import torch
    from _smi import sampled
    from mcgaze_amd import engine as E
    from mcgaze_amd.packing import bneck_stream
    g = torch.Generator().manual_seed(1)
    out = []
    for name, hw, cm, cn in SHAPES:
        w2 = torch.randn(cm, 3, 3, cm, generator=g) / (9 * cm / 2) ** 0.5
        w3 = torch.randn(4 * cm, cm, generator=g) / 8
        w1 = torch.randn(cn, 4 * cm, generator=g) / 11
        ws, bs = bneck_stream(w2, torch.randn(cm, generator=g) * 0.1, w3, torch.randn(4 * cm, generator=g) * 0.1, w1, torch.randn(cn, generator=g) * 0.1)
        ws, bs = ws.cuda(), bs.cuda()
        x = torch.randn(N, hw, hw, cm, device='cuda').relu_()
        res = torch.randn(N, hw, hw, 4 * cm, device='cuda').relu_()
        ms, clk, pw, n = sampled(lambda: E.bottleneck_x3(x, res, ws, bs, cn, 1), secs)
        out.append(dict(shape=name, ms=ms, sclk=clk, W=pw, J=ms * pw / 1e3, samples=n))
    print('RESULT ' + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        return child(int(sys.argv[2]), float(sys.argv[3]))
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 448
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    rows = {}
    for m in (0, 1, 2, 4, 8, 15, 0):
        env = dict(os.environ)
        lib = os.path.join(ROOT, 'mcgaze_amd', f'lab_bnx_{m}.so')
        if m:
            if not os.path.exists(lib):
                print(f'(no {lib}: run tools/lab/bneck_ablate.sh in the build container first)')
                continue
            env['MCGAZE_LIB'] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(N), str(secs)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
        if not line:
            print(f'variant {m} failed: {r.stderr[-400:]}')
            continue
        rows.setdefault(m, []).append(json.loads(line[0][7:]))
    base = rows.get(0)
    print(f'| shape ({N} frames) | variant | ms | sclk MHz | W | J / launch | ms vs product | J vs product |\n|---|---|---|---|---|---|---|---|')
    for i, (name, *_) in enumerate(SHAPES):
        b = base[0][i]
        for m, runs in rows.items():
            for k, run in enumerate(runs):
                r = run[i]
                tag = NAMES[m] + (' (second run, after the variants)' if (m == 0 and k == 1) else '')
                print(f"| {name} | {tag} | {r['ms']:.3f} | {r['sclk']:.0f} | {r['W']:.0f} | {r['J']:.3f} | {r['ms'] - b['ms']:+.3f} | {r['J'] - b['J']:+.3f} |")


if __name__ == '__main__':
    main()
