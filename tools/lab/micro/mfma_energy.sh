#!/bin/bash
# GPU: tools/lab/micro/mfma_energy.out <shape> <data mode> under rocm-smi sampling -> TFLOP/s, clock, watts per variant
R=$GRAFT_REPO_ROOT
for v in "32 0" "16 0" "32 2" "32 1" "32 0"; do
  set -- $v
  $R/tools/lab/micro/mfma_energy.out $1 $2 3.0 > /tmp/mfma_e.log 2>&1 &
  PID=$!
  sleep 1.2
  S=""
  for i in 1 2 3 4 5 6; do S="$S $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.2; done
  wait $PID
  echo "$(cat /tmp/mfma_e.log | grep mfma) | samples (sclk, W):$S"
done
