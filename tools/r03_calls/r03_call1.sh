set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/layer_profile.py 64 f16x3 > gpurun_out/r03_a_layers_x3.log 2>&1
python tools/layer_profile.py 64 bf16 > gpurun_out/r03_a_layers_bf16.log 2>&1
ITERS=1200 SLEEP=6 tools/power_probe.sh conv 50 f16x3 randn > gpurun_out/r03_a_power_x3_randn.log 2>&1
ITERS=1200 SLEEP=6 tools/power_probe.sh conv 50 f16x3 relu > gpurun_out/r03_a_power_x3_relu.log 2>&1
ITERS=3000 SLEEP=5 tools/power_probe.sh conv 14 bf16 randn > gpurun_out/r03_a_power_bf16_randn.log 2>&1
tail -3 gpurun_out/r03_a_power_*.log
tail -5 gpurun_out/r03_a_layers_x3.log
