set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_bottleneck or small_weights" -s > gpurun_out/r03_d_bneck_test.log 2>&1
grep -E "fused bottleneck|passed|failed|Error|error|f16x3 1x1" gpurun_out/r03_d_bneck_test.log | head -40
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "fused_bottleneck or batched_equals or golden" > gpurun_out/r03_d_fwd_test.log 2>&1
tail -15 gpurun_out/r03_d_fwd_test.log
timeout 300 python tools/layer_profile.py 64 f16x3 > gpurun_out/r03_d_layers_x3.log 2>&1
head -14 gpurun_out/r03_d_layers_x3.log; tail -2 gpurun_out/r03_d_layers_x3.log
timeout 600 python bench.py --steps 20 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 2>gpurun_out/r03_d_bench.err | tail -1 > gpurun_out/r03_d_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r03_d_bench.json')); print({k:d[k] for k in ('value','ms_per_step','verified','within_tolerance','max_abs_dev_yaw_pitch_clip0')})"
