cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03_r_gputest.log 2>&1
grep -E "passed|failed" gpurun_out/r03_r_gputest.log | tail -2
bash tools/decoder_prof.sh f16x3 2>&1 | grep -E "decoder only|pw_single_x3|igemm_dma_kernel<float, 256, 256"
for i in 1 2; do
python bench.py --steps 30 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['max_abs_dev_yaw_pitch_clip0'])"
done
