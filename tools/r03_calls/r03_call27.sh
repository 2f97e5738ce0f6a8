cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/micro/l2_per_cu.out 2>&1 | tee gpurun_out/r03_t_l2_per_cu.log
