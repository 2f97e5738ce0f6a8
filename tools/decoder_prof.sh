#!/bin/bash
# GPU: per-kernel time of the decoder alone (rocprofv3 kernel trace of tools/decoder_time.py)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
python $R/tools/decoder_time.py 20 ${1:-bf16}
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/dec_prof
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dec_prof -o trace -- python $R/tools/decoder_time.py 20 ${1:-bf16} > $R/gpurun_out/dec_prof.log 2>&1
DB=$(find $R/gpurun_out/dec_prof -name '*.db' | head -1)
cd $R && python tools/rocpd_summary.py "$DB" gpurun_out/dec_kernel_stats_${1:-bf16}.md | head -24
rm -rf gpurun_out/dec_prof
