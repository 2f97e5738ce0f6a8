cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for pr in -1 0; do
for prec in f16x3 bf16; do
python bench.py --steps 40 --warmup 5 --precision $prec --decoder-priority $pr --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none --host-input-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$prec priority $pr', d['value'], d['ms_per_step'])"
done; done; done
