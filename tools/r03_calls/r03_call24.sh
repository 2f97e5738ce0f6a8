cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 tools/micro/access_pattern.out 448 2>&1 | tee gpurun_out/r03_t_access_pattern.log
