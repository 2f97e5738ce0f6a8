cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
echo -n "new: "; timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k two_threads 2>&1 | grep -E "passed|failed" | tail -1
echo -n "base: "; MCGAZE_LIB=$PWD/mcgaze_amd/libmcgaze_hip_base.so timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k two_threads 2>&1 | grep -E "passed|failed" | tail -1
done
