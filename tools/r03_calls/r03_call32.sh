cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for lib in mcgaze_amd/libmcgaze_hip_base.so mcgaze_amd/libmcgaze_hip.so; do
echo -n "$lib: "
MCGAZE_LIB=$PWD/$lib python bench.py --steps 30 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['max_abs_dev_yaw_pitch_clip0'])"
done
done
python tools/decoder_time.py f16x3 2>&1 | tail -3
MCGAZE_LIB=$PWD/mcgaze_amd/libmcgaze_hip_base.so python tools/decoder_time.py f16x3 2>&1 | tail -3
