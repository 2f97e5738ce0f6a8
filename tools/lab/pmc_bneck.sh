#!/bin/bash
# GPU: PMC counters of the fused bottleneck tail alone.  usage: tools/lab/pmc_bneck.sh <tag> [bneck_bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o $n --output-format csv -- python $R/tools/bneck_bench.py $ARGS > $OUT/$n.log 2>&1; echo "$n rc=$?"; }
ARGS="$*"
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f'{out}/*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'bneck' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
for c, v in sorted(agg.items()):
    print(f'{c:28s} launches {len(v):3d}  mean {sum(v) / len(v):.5g}')
PY
