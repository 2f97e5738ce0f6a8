cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r03_h_bneck_lab.log
for lab in 0 1 2 3 4 7; do python tools/bneck_bench.py 448 56 56 1 64 30 64 $((lab*65536)) 1 >> gpurun_out/r03_h_bneck_lab.log 2>&1; done
for lab in 0 1 2 3 7; do python tools/bneck_bench.py 448 28 28 1 128 30 128 $((lab*65536)) >> gpurun_out/r03_h_bneck_lab.log 2>&1; done
grep -E "bneck_x3|tile 1" gpurun_out/r03_h_bneck_lab.log
