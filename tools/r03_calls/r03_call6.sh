set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_bottleneck" -s > gpurun_out/r03_f_bneck_test.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/r03_f_bneck_test.log | head; grep "cm=128" gpurun_out/r03_f_bneck_test.log | head -4
rm -f gpurun_out/r03_f_bneck_bench.log
for st in 0 4 8 14 20 30 50; do python tools/bneck_bench.py 448 56 56 1 64 50 64 $st >> gpurun_out/r03_f_bneck_bench.log 2>&1; done
for st in 0 14 30 60; do python tools/bneck_bench.py 448 28 28 1 128 50 128 $st >> gpurun_out/r03_f_bneck_bench.log 2>&1; done
python tools/bneck_bench.py 448 56 56 1 128 50 64 14 >> gpurun_out/r03_f_bneck_bench.log 2>&1
python tools/bneck_bench.py 448 56 56 2 64 50 64 14 >> gpurun_out/r03_f_bneck_bench.log 2>&1
python tools/bneck_bench.py 448 28 28 1 0 50 128 14 >> gpurun_out/r03_f_bneck_bench.log 2>&1
grep bneck_x3 gpurun_out/r03_f_bneck_bench.log
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "fused_bottleneck or batched_equals or golden" > gpurun_out/r03_f_fwd_test.log 2>&1
tail -4 gpurun_out/r03_f_fwd_test.log
timeout 300 python tools/layer_profile.py 64 f16x3 > gpurun_out/r03_f_layers_x3.log 2>&1
head -16 gpurun_out/r03_f_layers_x3.log; tail -1 gpurun_out/r03_f_layers_x3.log
timeout 600 python bench.py --steps 20 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 2>gpurun_out/r03_f_bench.err | tail -1 > gpurun_out/r03_f_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r03_f_bench.json')); print({k:d[k] for k in ('value','ms_per_step','verified','within_tolerance','max_abs_dev_yaw_pitch_clip0')})"
