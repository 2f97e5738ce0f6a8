#!/bin/bash
# GPU: clocks and power while a workload loops (is the matrix pipe power-limited?).
# usage: tools/lab/power_probe.sh conv [tile] [precision] [data]   -- the FPN P2 3x3 conv through one tile (f16x3: tile 50, ITERS=1500)
#        tools/lab/power_probe.sh bench          -- the whole path (bench.py, 600 steps)
R=$GRAFT_REPO_ROOT
if [ "${1:-conv}" = bench ]; then python $R/bench.py --steps 600 --warmup 5 --cpu-seconds 0 > /tmp/load.log 2>&1 &
else python $R/tools/conv_bench.py 448 56 56 256 256 3 1 1 ${ITERS:-4000} 0 ${2:-14} ${3:-bf16} ${4:-randn} > /tmp/load.log 2>&1 & fi
PID=$!
sleep ${SLEEP:-4}
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done
wait $PID; tail -1 /tmp/load.log | cut -c1-200
rocm-smi --showmaxpower 2>/dev/null | grep -i "max graphics"
