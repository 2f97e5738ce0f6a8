cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "x3_tile" 2>&1 | tail -2
python tools/layer_profile.py 1 f16x3 > gpurun_out/r03_v_layers_b1_tile53.log 2>&1
grep -E "^ *(11|14|29|31|37|41|42|43|44|48) |total" gpurun_out/r03_v_layers_b1_tile53.log
