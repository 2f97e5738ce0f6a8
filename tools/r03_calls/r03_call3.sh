set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/r03_lab_skip.py f16x3 20 > gpurun_out/r03_c_lab_skip_x3.log 2>&1
cat gpurun_out/r03_c_lab_skip_x3.log | grep skip
