#!/bin/bash
# GPU lab: HBM-side bytes and SQ wait / busy counters of the decoder's kernels (tools/decoder_time.py: 4 x [RoIAlign + stage] + gaze head on precomputed
# pyramids), per kernel symbol.  usage: tools/lab/pmc_decoder.sh [precision=f16x3]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_decoder; rm -rf $OUT; mkdir -p $OUT
P=${1:-f16x3}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/$N -o $N --output-format csv -- python $R/tools/decoder_time.py 6 $P > $OUT/$N.log 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{out}/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'].split('(')[0].replace('void ', '')[:60]][row['Counter_Name']].append(float(row['Counter_Value']))
print('| kernel | launches | fetch MB / launch (x 2) | write MB / launch | waves waiting | MFMA busy |')
print('|---|---|---|---|---|---|')
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_BUSY_CYCLES', [0]))):
    n = max(len(d.get('FETCH_SIZE', [])), 1)
    f = sum(d.get('FETCH_SIZE', [0])) / n * 1024 * 2 / 1e6
    w = sum(d.get('WRITE_SIZE', [0])) / max(len(d.get('WRITE_SIZE', [])), 1) * 1024 / 1e6
    busy = sum(d.get('SQ_BUSY_CYCLES', [0])); wc = sum(d.get('SQ_WAVE_CYCLES', [0]))
    wait = sum(d.get('SQ_WAIT_ANY', [0])) / wc if wc else 0
    mf = sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', [0])) / (busy / 32 * 1024) if busy else 0
    print(f'| `{k}` | {n} | {f:.1f} | {w:.1f} | {wait * 100:.0f} % | {mf * 100:.0f} % |')
PY
