cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/two_thread_probe.py 8 f16x3 full 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_x_two_thread_probe_fixed.log
timeout 1500 python tools/two_thread_probe.py 8 bf16 full 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_x_two_thread_probe_fixed.log
