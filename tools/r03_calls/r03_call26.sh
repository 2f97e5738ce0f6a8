cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r03_u_bneck.log
for a in "448 56 56 1 64 40 64 2" "448 28 28 1 128 40 128 2" ; do python tools/bneck_bench.py $a >> gpurun_out/r03_u_bneck.log 2>&1; done
grep -E "bneck_x3|tile [123]:|compute" gpurun_out/r03_u_bneck.log
