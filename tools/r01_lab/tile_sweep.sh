#!/bin/bash
# GPU: per-layer contraction timing for each DMA tile configuration (MCG_TILE); usage: tools/tile_sweep.sh "0 1 3 6 7"
TILES=${1:-"0 1 2 3 4 5"}
for t in $TILES; do
  MCG_TILE=$t python tools/layer_profile.py 64 bf16 > gpurun_out/layers_tile$t.log 2>&1
done
TILES="$TILES" python - <<'PY'
import os, re
tiles = os.environ['TILES'].split()
cols = []
for t in tiles:
    rows = [l.split() for l in open(f'gpurun_out/layers_tile{t}.log') if re.match(r'\s*\d+\s+\d+\s+\d+', l)]
    cols.append(rows)
print('idx        M      N      K | ms per MCG_TILE ' + ' '.join(tiles) + ' | best')
tot = [0.0] * len(tiles)
best_tot = 0.0
for i in range(len(cols[0])):
    ms = [float(c[i][5]) for c in cols]
    for t in range(len(tiles)): tot[t] += ms[t]
    b = min(range(len(tiles)), key=lambda t: ms[t]); best_tot += ms[b]
    r = cols[0][i]
    if i < 62: print(f'{i:3d} {r[2]:>8} {r[3]:>6} {r[4]:>6} | ' + ' '.join(f'{m:7.4f}' for m in ms) + f' | {tiles[b]}')
print('totals', ' '.join(f'{t:.3f}' for t in tot), 'best-per-layer total', f'{best_tot:.3f}')
PY
