set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for a in "448 56 56 1 64" "448 56 56 1 128" "448 56 56 2 64" "448 56 56 1 0" "224 56 56 1 64" "896 56 56 1 64"; do python tools/bneck_bench.py $a >> gpurun_out/r03_e_bneck_bench.log 2>&1; done
cat gpurun_out/r03_e_bneck_bench.log
tools/pmc_bneck.sh r03_e_bneck 448 56 56 1 64 20 > gpurun_out/r03_e_bneck_pmc.log 2>&1
cat gpurun_out/r03_e_bneck_pmc.log
