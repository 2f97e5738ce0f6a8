cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python tools/two_thread_probe.py 30 f16x3 full 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_x_two_thread_soak.log | tail -14
(timeout 900 python tools/bneck_contention_probe.py 64 64 20000 1; timeout 900 python tools/bneck_contention_probe.py 128 128 20000 1) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_x_two_thread_soak.log | tail -3
