#!/bin/bash
# GPU: SQ PMC counters for one conv shape.  usage: tools/lab/pmc_conv.sh <tag> <conv_bench args...>
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq -o sq --output-format csv -- python $R/tools/conv_bench.py "$@" > $OUT/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES -d $OUT/sq2 -o sq2 --output-format csv -- python $R/tools/conv_bench.py "$@" > $OUT/sq2.log 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ('sq', 'sq2'):
    files = glob.glob(f'{out}/{sub}/**/*counter_collection.csv', recursive=True)
    if not files:
        print(sub, 'no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        if 'igemm' in row['Kernel_Name']:
            agg[row['Kernel_Name'][:70]][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, d in agg.items():
        print(sub, k, {c: f'{sum(v) / len(v):.4g}' for c, v in d.items()})
PY
