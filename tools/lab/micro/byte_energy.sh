#!/bin/bash
# GPU lab: tools/lab/micro/byte_energy.out <mode> under rocm-smi sampling -> GB/s, shader clock, watts per mode, and pJ per byte against the
# no-memory mode of the same grid (mode 4).  usage: tools/lab/micro/byte_energy.sh
R=$GRAFT_REPO_ROOT
declare -A W RATE
for m in 4 0 1 2 3 4; do
  $R/tools/lab/micro/byte_energy.out $m 5.0 > /tmp/byte_e.log 2>&1 &
  PID=$!
  sleep 1.5
  S=""; SUM=0; N=0
  for i in 1 2 3 4 5 6; do
    L=$(rocm-smi --showclocks --showpower 2>/dev/null)
    CLK=$(echo "$L" | grep -oE 'sclk clock level: \S+ \(([0-9]+)Mhz\)' | grep -oE '[0-9]+Mhz' | head -1)
    P=$(echo "$L" | grep -oE 'Power \(W\): [0-9.]+' | grep -oE '[0-9.]+$' | head -1)
    S="$S ($CLK, $P W)"
    if [ -n "$P" ]; then SUM=$(python3 -c "print($SUM + $P)"); N=$((N+1)); fi
    sleep 0.3
  done
  wait $PID
  AVG=$(python3 -c "print(round($SUM / max($N, 1), 1))")
  RT=$(grep -oE '[0-9]+ GB/s' /tmp/byte_e.log | grep -oE '[0-9]+')
  echo "$(cat /tmp/byte_e.log) | mean $AVG W | samples:$S"
  W[$m]=$AVG; RATE[$m]=$RT
done
python3 - <<PY
w = {0: ${W[0]}, 1: ${W[1]}, 2: ${W[2]}, 3: ${W[3]}, 4: ${W[4]}}
r = {0: ${RATE[0]:-0}, 1: ${RATE[1]:-0}, 2: ${RATE[2]:-0}, 3: ${RATE[3]:-0}}
names = {0: 'HBM read', 1: 'HBM write', 2: 'HBM copy (read + write)', 3: 'L2-resident read'}
print('| stream | GB/s | package W | W above the no-memory grid (%.0f W) | pJ per byte |' % w[4])
print('|---|---|---|---|---|')
for m in (0, 1, 2, 3):
    print('| %s | %d | %.0f | %.0f | %.0f |' % (names[m], r[m], w[m], w[m] - w[4], (w[m] - w[4]) / max(r[m], 1) * 1e3))
PY
