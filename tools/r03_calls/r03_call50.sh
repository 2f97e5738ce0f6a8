cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_bottleneck_tail_f16x3" 2>&1 | tail -2
