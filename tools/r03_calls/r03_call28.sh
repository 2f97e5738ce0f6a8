cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r03_u_ab.log
for rep in 1 2; do
for d in ab_base .; do
for a in "448 56 56 1 64 40 64" "448 56 56 1 128 40 64" "448 56 56 2 64 40 64" "448 56 56 1 0 40 64" "448 28 28 1 128 40 128" "448 28 28 1 0 40 128"; do (cd $d; echo -n "$d: "; python tools/bneck_bench.py $a 2>&1 | grep bneck_x3) >> gpurun_out/r03_u_ab.log; done
done
done
cat gpurun_out/r03_u_ab.log
