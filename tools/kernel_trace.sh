#!/bin/bash
# GPU: rocprofv3 kernel-trace summary of an arbitrary command.  usage: tools/kernel_trace.sh <tag> <command...>  -> gpurun_out/<tag>_kernel_stats.md
TAG=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o trace -- "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_prof -name '*.db' | head -1)
cd $R && python tools/rocpd_summary.py "$DB" gpurun_out/${TAG}_kernel_stats.md | head -${LINES_OUT:-16}
rm -rf gpurun_out/${TAG}_prof
