cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/x3_tile_sweep.py 30 > gpurun_out/r03_j_x3_tile_sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/r03_j_x3_tile_sweep.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "f16x3 and not bottleneck" 2>&1 | tail -2
