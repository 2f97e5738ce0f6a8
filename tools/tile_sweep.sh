#!/bin/bash
# GPU: per-layer contraction timing for each DMA tile configuration (MCG_TILE)
for t in 0 1 2 3 4 5; do
  MCG_TILE=$t python tools/layer_profile.py 64 bf16 > gpurun_out/layers_tile$t.log 2>&1
done
python - <<'PY'
import re
cols = []
for t in range(6):
    rows = [l.split() for l in open(f'gpurun_out/layers_tile{t}.log') if re.match(r'\s*\d+\s+\d+\s+\d+', l)]
    cols.append(rows)
print('idx        M      N      K | ms per MCG_TILE 0..5 (0=128x128x64B/4w/S4 1=256x128/4w 2=256x128/8w 3=256x256/8w 4=128x128x128B/S3 5=256x128x128B/8w) | best')
tot = [0.0] * 6
best_tot = 0.0
for i in range(len(cols[0])):
    ms = [float(c[i][5]) for c in cols]
    for t in range(6): tot[t] += ms[t]
    b = min(range(6), key=lambda t: ms[t]); best_tot += ms[b]
    r = cols[0][i]
    if i < 62: print(f'{i:3d} {r[2]:>8} {r[3]:>6} {r[4]:>6} | ' + ' '.join(f'{m:7.4f}' for m in ms) + f' | {b}')
print('totals', ' '.join(f'{t:.3f}' for t in tot), 'best-per-layer total', f'{best_tot:.3f}')
PY
