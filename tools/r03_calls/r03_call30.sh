cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "x3_tile or f16x3" 2>&1 | tail -2
python tools/layer_profile.py 1 f16x3 > gpurun_out/r03_v_layers_b1_tile53.log 2>&1
grep -E "^ *(11|14|29|31|37|41|42|43|44|48) |total" gpurun_out/r03_v_layers_b1_tile53.log
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "batched_equals or golden or graphed or bench_schedule" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --second-engine none --cpu-seconds 0 --mae-videos 0 --backbone-clips 0 --kernel-events none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['max_abs_dev_yaw_pitch_clip0']); print(d['latency_single_clip'])"
