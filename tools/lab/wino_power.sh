#!/bin/bash
# GPU: clocks and package power while wino_bench.py alternates the direct f16x3 kernel and wino_x3 on the FPN P2 shape
R=$GRAFT_REPO_ROOT
python $R/tools/wino_bench.py 448 56 56 256 256 ${ITERS:-700} ${1:-randn} > /tmp/load.log 2>&1 &
PID=$!
sleep 6
for i in $(seq 1 ${SAMPLES:-26}); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.4; done
wait $PID; grep -v amdgpu.ids /tmp/load.log
