cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --second-engine none --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --kernel-events none 2>gpurun_out/r03_host.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified']); h=d['host_input']; print(h['f32'], h['uint8'])"
tail -5 gpurun_out/r03_host.err
