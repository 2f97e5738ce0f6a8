cd $GRAFT_REPO_ROOT
for w in full backbone_fpn full backbone_fpn; do
python bench.py --steps 30 --warmup 5 --workload $w --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none --host-input-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])"
done
python tools/decoder_time.py 20 f16x3 2>&1 | tail -3
