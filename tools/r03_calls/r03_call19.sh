cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "pointwise_stream_x3" 2>&1 | tail -2
timeout 300 python tools/layer_profile.py 64 f16x3 > gpurun_out/r03_o_layers_x3.log 2>&1
grep -E "^ *[0-9]+ +71 " gpurun_out/r03_o_layers_x3.log | tail -3; tail -1 gpurun_out/r03_o_layers_x3.log
