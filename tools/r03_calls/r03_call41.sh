cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_bottleneck" 2>&1 | tail -2
for rep in 1 2; do
for v in _base ""; do
export MCGAZE_LIB=$PWD/mcgaze_amd/libmcgaze_hip$v.so
echo "== lib$v"
for a in "448 56 56 1 64 40 64" "448 56 56 1 128 40 64" "448 56 56 2 64 40 64" "448 28 28 1 128 40 128" "448 28 28 1 0 40 128"; do python tools/bneck_bench.py $a 2>&1 | grep bneck_x3; done
python bench.py --steps 30 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['max_abs_dev_yaw_pitch_clip0'])"
done
done
unset MCGAZE_LIB
(timeout 600 python tools/bneck_contention_probe.py 64 128 3000 1; timeout 600 python tools/bneck_contention_probe.py 128 128 3000 1) 2>&1 | grep -v amdgpu.ids | tail -4
