cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "pointwise_stream_x3 or golden or batched_equals" 2>&1 | tail -12
timeout 300 python tools/layer_profile.py 64 f16x3 > gpurun_out/r03_o_layers_x3.log 2>&1
grep -E "^ *[0-9]+ +71 " gpurun_out/r03_o_layers_x3.log; tail -1 gpurun_out/r03_o_layers_x3.log
for i in 1 2; do
python bench.py --steps 30 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['max_abs_dev_yaw_pitch_clip0'])"
done
