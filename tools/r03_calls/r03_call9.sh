cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r03_i_bneck.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_bottleneck" > gpurun_out/r03_i_bneck_test.log 2>&1
tail -2 gpurun_out/r03_i_bneck_test.log
python tools/bneck_bench.py 448 56 56 1 64 30 64 0 1 >> gpurun_out/r03_i_bneck.log 2>&1
python tools/bneck_bench.py 448 56 56 1 128 30 64 0 >> gpurun_out/r03_i_bneck.log 2>&1
python tools/bneck_bench.py 448 56 56 2 64 30 64 0 >> gpurun_out/r03_i_bneck.log 2>&1
python tools/bneck_bench.py 448 28 28 1 128 30 128 0 1 >> gpurun_out/r03_i_bneck.log 2>&1
python tools/bneck_bench.py 448 28 28 1 0 30 128 0 >> gpurun_out/r03_i_bneck.log 2>&1
grep -E "bneck_x3|tile [12]" gpurun_out/r03_i_bneck.log
