#!/bin/bash
# GPU: matrix-pipe utilisation per kernel symbol over single-stream steps of bench.py: SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32
# shader engines x 1024 SIMDs), summed over every launch of the symbol.  Prints a markdown table (-> profiles/).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_mfma; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/sq -o sq --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --power-seconds 0 --kernel-events none --pipeline 0 --trunk-streams 1 --second-engine none --exact-steps 0 --latency 0 --mae-videos 0 --backbone-clips 0 --precision ${PRECISION:-f16x3} > $OUT/sq.log 2>&1
echo "pass rc=$?"
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.join(os.getcwd(), 'tools')); sys.path.insert(0, os.getcwd())
from _symbols import cfg_name_of
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(f'{out}/sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        if row['Counter_Name'] == 'SQ_BUSY_CYCLES': calls[k] += 1
tot_busy = sum(d['SQ_BUSY_CYCLES'] for d in agg.values())
tot_mfma = sum(d['SQ_VALU_MFMA_BUSY_CYCLES'] for d in agg.values())
print('| kernel | launches | share of GPU cycles | matrix pipe busy | waves waiting (WAIT_ANY / WAVE_CYCLES) |')
print('|---|---|---|---|---|')
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_BUSY_CYCLES']):
    if d['SQ_BUSY_CYCLES'] / tot_busy < 0.004: continue
    util = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['SQ_BUSY_CYCLES'] / 32 * 1024)
    print(f"| `{k[:70]}` | {calls[k]} | {d['SQ_BUSY_CYCLES'] / tot_busy * 100:.1f} % | {util * 100:.1f} % | {d['SQ_WAIT_ANY'] / max(d['SQ_WAVE_CYCLES'], 1) * 100:.0f} % |")
print(f'| all kernels | {sum(calls.values())} | 100 % | {tot_mfma / (tot_busy / 32 * 1024) * 100:.1f} % | |')
# the same per bench.py roofline name (several template instances share one), for roofline.top_symbols: gpurun_out/pmc_mfma.json -> profiles/
by = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f'{out}/sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        by[cfg_name_of(row['Kernel_Name'])][row['Counter_Name']] += float(row['Counter_Value'])
from mcgaze_amd import lib as L
js = {n: {'mfma_busy': round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['SQ_BUSY_CYCLES'] / 32 * 1024), 4), 'waves_waiting': round(d['SQ_WAIT_ANY'] / max(d['SQ_WAVE_CYCLES'], 1), 4)}
      for n, d in by.items() if d['SQ_BUSY_CYCLES'] / tot_busy >= 0.004}
js['_all_kernels_mfma_busy'] = round(tot_mfma / (tot_busy / 32 * 1024), 4)
js['_build_id'] = L.build_id()
js['_how'] = 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY over single-stream bench.py steps (tools/pmc_bench_mfma.sh); busy = MFMA_BUSY / (BUSY / 32 x 1024 SIMDs)'
json.dump(js, open(f'{out}/../pmc_mfma.json', 'w'), indent=1)
PY
