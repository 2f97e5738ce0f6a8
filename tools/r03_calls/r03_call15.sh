cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
PRECISION=f16x3 bash tools/round_profile.sh r03_m_x3 > gpurun_out/r03_m_x3_round.log 2>&1
tail -20 gpurun_out/r03_m_x3_round.log
PRECISION=f16x3 bash tools/pmc_bench_mfma.sh > gpurun_out/r03_m_x3_mfma_util.md 2>gpurun_out/r03_m_x3_mfma.err
cat gpurun_out/r03_m_x3_mfma_util.md
