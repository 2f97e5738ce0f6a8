cd $GRAFT_REPO_ROOT
bash tools/round_profile.sh r03_y > gpurun_out/r03_y_round_profile.log 2>&1
tail -20 gpurun_out/r03_y_round_profile.log
tools/pmc_bench_mfma.sh > gpurun_out/r03_y_mfma.log 2>&1; tail -12 gpurun_out/r03_y_mfma.log
