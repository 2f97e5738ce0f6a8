cd $GRAFT_REPO_ROOT
for ts in 2 3 4 2 3; do
python bench.py --steps 30 --warmup 5 --trunk-streams $ts --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none --host-input-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('trunk_streams $ts', d['value'], d['ms_per_step'], d['verified'])"
done
