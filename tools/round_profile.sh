#!/bin/bash
# GPU: the evidence bundle of one build -- bench line, rocprofv3 kernel-trace summary of the same command, PMC HBM traffic.
# usage: tools/round_profile.sh <tag>   -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats.md, pmc_traffic.json
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
python $R/bench.py --steps 20 --warmup 5 2> $R/gpurun_out/${TAG}_bench.err | tail -1 > $R/gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o trace -- python $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --power-seconds 0 --second-engine none --exact-steps 0 --latency 0 --mae-videos 0 --backbone-clips 0 --precision ${PRECISION:-f16x3} > $R/gpurun_out/${TAG}_prof.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_prof -name '*.db' | head -1)
cd $R
python tools/rocpd_summary.py "$DB" gpurun_out/${TAG}_kernel_stats.md > /dev/null
rm -rf gpurun_out/${TAG}_prof
# the same trace with nothing concurrent (one trunk stream, no batch pipeline): the per-launch durations bench.py's roofline uses
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof1 -o trace -- python $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --power-seconds 0 --pipeline 0 --trunk-streams 1 --second-engine none --exact-steps 0 --latency 0 --mae-videos 0 --backbone-clips 0 --precision ${PRECISION:-f16x3} > $R/gpurun_out/${TAG}_prof1.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_prof1 -name '*.db' | head -1)
cd $R
python tools/rocpd_summary.py "$DB" gpurun_out/${TAG}_kernel_stats_serial.md > /dev/null
rm -rf gpurun_out/${TAG}_prof1
PRECISION=${PRECISION:-f16x3} tools/pmc_bench_traffic.sh > gpurun_out/${TAG}_pmc.log 2>&1
rm -rf gpurun_out/pmc_bench
cat gpurun_out/${TAG}_bench.json; head -8 gpurun_out/${TAG}_kernel_stats.md; head -6 gpurun_out/${TAG}_kernel_stats_serial.md
