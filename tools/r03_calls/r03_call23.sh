cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r03_s_bneck.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_bottleneck" 2>&1 | tail -1
for a in "448 56 56 1 64 40 64 1" "448 56 56 1 128 40 64" "448 56 56 2 64 40 64" "448 28 28 1 128 40 128 1" "448 28 28 1 0 40 128"; do python tools/bneck_bench.py $a >> gpurun_out/r03_s_bneck.log 2>&1; done
grep -E "bneck_x3|tile [12]:" gpurun_out/r03_s_bneck.log
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "fused_bottleneck or batched_equals or golden or full_batch" 2>&1 | tail -1
for i in 1 2; do
python bench.py --steps 30 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['max_abs_dev_yaw_pitch_clip0'])"
done
