set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r03_g_bneck_bench.log
for st in 0 14 60 200; do python tools/bneck_bench.py 448 56 56 1 64 50 64 $st >> gpurun_out/r03_g_bneck_bench.log 2>&1; done
python tools/bneck_bench.py 448 56 56 1 64 30 64 0 1 >> gpurun_out/r03_g_bneck_bench.log 2>&1
python tools/bneck_bench.py 448 28 28 1 128 30 128 0 1 >> gpurun_out/r03_g_bneck_bench.log 2>&1
python tools/bneck_bench.py 448 28 28 1 0 30 128 0 1 >> gpurun_out/r03_g_bneck_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r03_g_bneck_bench.log
