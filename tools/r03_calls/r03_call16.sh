cd $GRAFT_REPO_ROOT
for pr in -1 0 -1 0; do
python bench.py --steps 30 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none --decoder-priority $pr 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prio', $pr, d['value'], d['ms_per_step'], d['verified'])"
done
for ts in 1 3; do
python bench.py --steps 30 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none --trunk-streams $ts 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('trunk_streams', $ts, d['value'], d['ms_per_step'], d['verified'])"
done
