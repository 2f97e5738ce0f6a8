cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in orig slack both; do
export MCGAZE_LIB=$PWD/mcgaze_amd/libmcgaze_hip_$v.so
echo "== $v"
for a in "448 56 56 1 64 40 64" "448 56 56 1 128 40 64" "448 28 28 1 128 40 128"; do python tools/bneck_bench.py $a 2>&1 | grep bneck_x3; done
python bench.py --steps 30 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['max_abs_dev_yaw_pitch_clip0'])"
done
done
