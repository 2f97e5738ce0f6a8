cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "host_input" 2>&1 | tail -3
