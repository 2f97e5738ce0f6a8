#!/bin/bash
# GPU tool: A/B an engine option through bench.py, alternating values on the same box.  usage: tools/ab_option.sh NAME V1 V2 [rounds=2]
name=$1; v1=$2; v2=$3; rounds=${4:-2}
for r in $(seq 1 $rounds); do
  for v in $v1 $v2; do
    python bench.py --steps 20 --warmup 5 --exact-steps 0 --engine-option $name=$v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name=$v', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done
