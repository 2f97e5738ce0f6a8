cd $GRAFT_REPO_ROOT
for pr in -1 0 -1 0; do
python bench.py --steps 30 --warmup 5 --decoder-priority $pr --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none --host-input-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('decoder priority $pr', d['value'], d['ms_per_step'], d['verified'])"
done
