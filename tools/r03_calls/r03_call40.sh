cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "soak or two_threads" 2>&1 | tail -3
