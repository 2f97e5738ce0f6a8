#!/bin/bash
# GPU: HBM traffic counters (separate passes, each under its own timeout) for one conv shape.
# usage: tools/lab/pmc_traffic.sh <tag> <conv_bench args...>
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/traffic_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o $C --output-format csv -- python $R/tools/conv_bench.py "$@" 3 > $OUT/$C.log 2>&1
  echo "$C pass rc=$?"
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(f'{out}/{sub}/**/*counter_collection.csv', recursive=True)
    if not files:
        print(sub, 'no csv'); continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if 'igemm' in row['Kernel_Name']:
            agg[(row['Kernel_Name'][:64], row['Counter_Name'])].append(float(row['Counter_Value']))
    for k, v in agg.items():
        print(sub, k, 'avg per dispatch (KiB units):', round(sum(v) / len(v)), 'dispatches', len(v))
PY
