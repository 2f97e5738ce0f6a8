cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python tools/parity_fuzz.py 400 2 f16x3 > gpurun_out/r03_z_fuzz_400_s2.md 2>gpurun_out/r03_z_fuzz_a.err; tail -3 gpurun_out/r03_z_fuzz_400_s2.md) &
(timeout 2700 python tools/parity_fuzz.py 1000 7 f16x3 > gpurun_out/r03_z_fuzz_1000_s7.md 2>gpurun_out/r03_z_fuzz_b.err; tail -3 gpurun_out/r03_z_fuzz_1000_s7.md) &
(timeout 2700 python tools/parity_fuzz.py 1000 11 f16x3 > gpurun_out/r03_z_fuzz_1000_s11.md 2>gpurun_out/r03_z_fuzz_c.err; tail -3 gpurun_out/r03_z_fuzz_1000_s11.md) &
wait
