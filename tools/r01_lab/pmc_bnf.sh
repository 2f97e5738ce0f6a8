#!/bin/bash
# GPU: SQ PMC counters + kernel time of the fused bottleneck kernel (MCG_FUSED_BLOCK=1).  usage: tools/pmc_bnf.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_bnf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MCG_FUSED_BLOCK=1 MCG_TRUNK_STREAMS=1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $R/tools/trunk_time.py > $OUT/kt.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq -o sq --output-format csv -- python $R/tools/trunk_time.py > $OUT/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES -d $OUT/sq2 -o sq2 --output-format csv -- python $R/tools/trunk_time.py > $OUT/sq2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT -d $OUT/sq3 -o sq3 --output-format csv -- python $R/tools/trunk_time.py > $OUT/sq3.log 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(f'{out}/kt/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row['Name'] for k in ('bottleneck', 'c64', 'igemm')):
            print(row['Name'][:60], row['Calls'], row['AverageNs'], row['TotalDurationNs'])
for sub in ('sq', 'sq2', 'sq3'):
    files = glob.glob(f'{out}/{sub}/**/*counter_collection.csv', recursive=True)
    if not files:
        print(sub, 'no csv'); print(open(f'{out}/{sub}.log').read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        if 'bottleneck' in row['Kernel_Name']:
            agg[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, d in agg.items():
        print(sub, k, {c: f'{sum(v) / len(v):.4g}' for c, v in d.items()})
PY
