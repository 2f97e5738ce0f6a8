cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python tools/bneck_contention_probe.py 64 64 6000 1; timeout 600 python tools/bneck_contention_probe.py 128 128 6000 1; timeout 600 python tools/bneck_contention_probe.py 64 128 4000 1;  timeout 600 python tools/bneck_contention_probe.py 128 0 4000 1) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_x_bneck_contention_fixed.log | tail -8
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k two_threads 2>&1 | grep -E "passed|failed" | tail -1; done
for a in "448 56 56 1 64 40 64" "448 56 56 1 128 40 64" "448 56 56 2 64 40 64" "448 28 28 1 128 40 128" "448 28 28 1 0 40 128"; do python tools/bneck_bench.py $a 2>&1 | grep bneck_x3; done
